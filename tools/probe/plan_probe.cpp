// Prints stage / sub-stage counts of the planner for a spin-layout ansatz (planning experiments).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../aqc_research_amd/csrc/aqc_plan.h"
using namespace aqc;
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 16, L = argc > 2 ? atoi(argv[2]) : 40, colbits = argc > 3 ? atoi(argv[3]) : 0;
    int cyclic = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<int32_t> blocks(2 * L);
    std::vector<std::pair<int,int>> pairs;
    if (!cyclic) { for (int s = 0; s < 2; ++s) for (int i = s; i < n - 1; i += 2) pairs.push_back({i, i + 1}); }
    for (int i = 0; i < L; ++i) {
        std::pair<int,int> p;
        if (cyclic) { int off = (n % 2 == 0) ? (i / (n / 2)) % 2 : 0; p = {(2 * i + off) % n, (2 * i + off + 1) % n}; }
        else p = pairs[i % pairs.size()];
        blocks[i] = p.first; blocks[L + i] = p.second;
    }
    Program prog;
    std::string e = build_program(n, 0, blocks.data(), L, false, false, prog);
    if (!e.empty()) { printf("err %s\n", e.c_str()); return 1; }
    for (int inv = 0; inv < 2; ++inv)
    for (int k = 10; k <= 13; ++k) for (int lb = 3; lb >= 0; --lb) for (int mo : {1 << 30}) {
        Plan p = make_plan(prog, colbits, k, lb, inv);
        split_substages(prog, p, 4, mo);
        int ns = 0; std::string detail;
        for (auto& st : p.stages) { ns += st.subs.size(); detail += " " + std::to_string(st.ops.size()) + "/" + std::to_string(st.subs.size()); }
        printf("inv=%d k=%d low=%d maxops=%d stages=%zu subs=%d  [%s ]", inv, k, lb, mo, p.stages.size(), ns, detail.c_str());
        for (auto& st : p.stages) { printf(" {"); for (int b : st.bits) printf("%d,", b); printf("}"); }
        printf("\n");
    }
}
