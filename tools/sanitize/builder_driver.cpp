// builder_driver.cpp -- drives the threaded k-d builder (nt_kdtree_build, csrc/nt_builder.cpp) for the sanitizer runs of
// tools/sanitize.sh: a cloud of small random simplices, big enough for subtrees and node scans to go to worker threads;
// checks the tree's shape (every item in at least one leaf, indices in range) and frees it.  argv[1] = items (default 6000).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../include/ntracer_hip.h"

int main(int argc, char **argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 6000;
    int bad = 0;
    for (int n : {3, 4, 6}) {
        std::mt19937 rng(1234 + n);
        std::uniform_real_distribution<float> pos(-10.0f, 10.0f), off(-0.4f, 0.4f);
        std::vector<float> lo((size_t)items * n), hi((size_t)items * n), verts((size_t)items * n * n);
        std::vector<int32_t> first(items + 1);
        for (int i = 0; i < items; ++i) {
            first[i] = i;
            float c[16];
            for (int k = 0; k < n; ++k) c[k] = pos(rng);
            for (int k = 0; k < n; ++k) { lo[(size_t)i * n + k] = 1e30f; hi[(size_t)i * n + k] = -1e30f; }
            for (int v = 0; v < n; ++v)
                for (int k = 0; k < n; ++k) {
                    const float x = c[k] + off(rng);
                    verts[((size_t)i * n + v) * n + k] = x;
                    if (x < lo[(size_t)i * n + k]) lo[(size_t)i * n + k] = x;
                    if (x > hi[(size_t)i * n + k]) hi[(size_t)i * n + k] = x;
                }
        }
        first[items] = items;
        nt_kdtree t{};
        nt_kdtree_params p{};
        const int r = nt_kdtree_build(n, items, lo.data(), hi.data(), first.data(), verts.data(), &p, &t);
        if (r != NT_OK) { fprintf(stderr, "n=%d: nt_kdtree_build -> %d (%s)\n", n, r, nt_last_error()); return 2; }
        std::vector<char> seen(items, 0);
        for (int i = 0; i < t.n_nodes; ++i) {
            if (t.node_axis[i] >= 0) {
                if (t.node_left[i] < -1 || t.node_left[i] >= t.n_nodes || t.node_right[i] < -1 || t.node_right[i] >= t.n_nodes) ++bad;
            } else {
                const int st = t.node_left[i], cnt = t.node_right[i];
                if (st < 0 || cnt < 1 || st + cnt > t.n_leaf_items) { ++bad; continue; }
                for (int k = st; k < st + cnt; ++k) {
                    if (t.leaf_items[k] < 0 || t.leaf_items[k] >= items) ++bad; else seen[t.leaf_items[k]] = 1;
                }
            }
        }
        int missing = 0;
        for (char s : seen) missing += s ? 0 : 1;
        printf("n=%d: %d items -> %d nodes, %d leaf entries, %d items in no leaf, %d bad indices\n", n, items, t.n_nodes, t.n_leaf_items, missing, bad);
        bad += missing;
        nt_kdtree_free(&t);
    }
    return bad ? 1 : 0;
}
