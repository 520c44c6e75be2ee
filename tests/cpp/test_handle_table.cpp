// CPU-only unit test of the engine's handle table (csrc/runtime.hpp: HandleTable): pages of 4096 slots indexed by the handle, allocated
// with the first handle of a page and freed with its last one.  Built and run by tests/test_handle_table_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include "runtime.hpp"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    using fm::HandleTable; using fm::Node;
    HandleTable t;
    CHECK(t.size() == 0 && t.get(0) == nullptr && t.get(-5) == nullptr && t.get(1) == nullptr && t.get(int64_t(1) << 40) == nullptr);
    t.erase(7); t.erase(-1);                                        // nothing there: no effect
    std::vector<Node> nodes(20000);
    std::unordered_map<int64_t, Node*> model;
    std::mt19937_64 rng(12345);
    int64_t next = 1;
    for (int round = 0; round < 200000; ++round) {
        const int what = (int)(rng() % 10);
        if (what < 5 && model.size() < nodes.size()) {              // hand out the next handle (ids only grow; sometimes with a gap, like reserved ranges)
            if (rng() % 50 == 0) next += (int64_t)(rng() % 20000);
            Node* nd = &nodes[model.size()];
            // (the slot in `nodes` is arbitrary: the table only stores the pointer)
            nd = &nodes[(size_t)(rng() % nodes.size())];
            t.put(next, nd); model[next] = nd; ++next;
        } else if (!model.empty()) {                                // release a random live handle
            auto it = model.begin();
            std::advance(it, (long)(rng() % std::min<size_t>(model.size(), 64)));
            t.erase(it->first);
            CHECK(t.get(it->first) == nullptr);
            model.erase(it);
        }
        if (round % 997 == 0) {
            CHECK(t.size() == model.size());
            for (const auto& kv : model) CHECK(t.get(kv.first) == kv.second);
            size_t seen = 0;
            t.for_each([&](Node*) { ++seen; });
            CHECK(seen == model.size());
            CHECK(t.get(next) == nullptr && t.get(next + 4096) == nullptr);
        }
    }
    // overwrite keeps the count; clear empties
    if (!model.empty()) { const int64_t id = model.begin()->first; t.put(id, &nodes[0]); CHECK(t.size() == model.size() && t.get(id) == &nodes[0]); }
    t.clear();
    CHECK(t.size() == 0);
    for (const auto& kv : model) CHECK(t.get(kv.first) == nullptr);
    std::printf("handle table ok\n");
    return 0;
}
