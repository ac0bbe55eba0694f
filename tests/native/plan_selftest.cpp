// Sanitizer self-test of the host-only stage planner (aqc_plan.cpp; built with -fsanitize=address,undefined by
// tests/test_native_sanitizers.py): many random programs x tile sizes x low bits x directions x column bits,
// every plan and every sub-stage split must pass check_plan.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../aqc_research_amd/csrc/aqc_plan.h"

using namespace aqc;

int main() {
    srand(11);
    int failures = 0, plans = 0;
    for (int n = 2; n <= 14; ++n)
        for (int rep = 0; rep < 6; ++rep) {
            const int ent = rand() % 3;
            const bool trotter = ent == 0 && rep % 3 == 2;
            int L;
            std::vector<int32_t> blocks;
            if (trotter) {
                const int layers = 1 + rep % 2;
                std::vector<int> pairs;
                for (int q = 0; q + 1 < n; q += 2) pairs.push_back(q);
                for (int q = 1; q + 1 < n; q += 2) pairs.push_back(q);
                L = 3 * (n - 1) * layers;
                blocks.assign(2 * (size_t)L, 0);
                int i = 0;
                for (int l = 0; l < layers; ++l)
                    for (int q : pairs)
                        for (int k = 0; k < 3; ++k, ++i) {
                            blocks[i] = (k == 1) ? q : q + 1;
                            blocks[L + i] = (k == 1) ? q + 1 : q;
                        }
            } else {
                L = rand() % 40;
                blocks.assign(2 * (size_t)L, 0);
                for (int i = 0; i < L; ++i) {
                    blocks[i] = rand() % n;
                    do { blocks[L + i] = rand() % n; } while (blocks[L + i] == blocks[i]);
                }
            }
            Program prog;
            const std::string err = build_program(n, ent, blocks.data(), L, trotter, trotter && n % 2 == 0 && rep % 2 == 0, prog);
            if (!err.empty()) { printf("build_program(n=%d, L=%d, trotter=%d): %s\n", n, L, (int)trotter, err.c_str()); ++failures; continue; }
            for (int col_bits = 0; col_bits <= 3; col_bits += 3)
                for (int k = 2; k <= 13; ++k)
                    for (int low = 0; low <= 3; ++low)
                        for (int inv = 0; inv < 2; ++inv) {
                            Plan plan = make_plan(prog, col_bits, k, low, inv != 0);
                            std::string e = check_plan(prog, plan);
                            if (e.empty() && (int)plan.stages.front().bits.size() >= 3) {
                                for (int r = 3; r <= 4 && e.empty(); ++r) {
                                    if ((int)plan.stages.front().bits.size() < r) continue;
                                    Plan p2 = plan;
                                    split_substages(prog, p2, r, ent == 2 ? 4 : 8);
                                    e = check_plan(prog, p2);
                                }
                            }
                            ++plans;
                            if (!e.empty()) { printf("n=%d L=%d ent=%d cols=%d k=%d low=%d inv=%d: %s\n", n, L, ent, col_bits, k, low, inv, e.c_str()); ++failures; }
                        }
        }
    printf("%d plans checked, %d failures\n", plans, failures);
    return failures != 0;
}
