// Single-threaded rehearsal of the device SAH build's passes (pbrt-v3-rs_amd/csrc/bvh_sah_steps.h) against the host builder (bvh_build.cpp), on the CPU:
// the same step functions the kernels run, called in grid order by plain loops.  A development aid for the step logic only — the kernels themselves are
// tested on the GPU (tests/test_bvh_device_gpu.py).  Build and run:  bash scripts/sah_steps_check.sh
#include "../pbrt-v3-rs_amd/csrc/bvh_sah_steps.h"
#include "../pbrt-v3-rs_amd/csrc/bvh_build.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

using namespace phs;

static int run(size_t n, unsigned seed, int max_prims, int mode, int forest = 0) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    std::vector<float> P; std::vector<uint32_t> idx;
    for (size_t t = 0; t < n; t++) {
        float c[3] = {U(rng), U(rng), U(rng)};
        if (mode == 1) { c[0] = std::floor(c[0] * 3.0f); c[1] = std::floor(c[1] * 3.0f); c[2] = 0.0f; }   // many coincident centroids, a flat scene
        if (mode == 2 && (t & 3)) { c[0] = 0.25f; c[1] = -0.5f; c[2] = 0.75f; }                               // three quarters of the triangles in one place
        const float s = mode == 1 ? 0.5f : 0.05f;
        for (int v = 0; v < 3; v++) {
            float d[3] = {U(rng) * s, U(rng) * s, U(rng) * s};
            if (mode == 1 || (mode == 2 && (t & 3))) { d[0] = (v == 1) * s; d[1] = (v == 2) * s; d[2] = 0.0f; }
            for (int k = 0; k < 3; k++) P.push_back(c[k] + d[k]);
            idx.push_back((uint32_t)(3 * t + v));
        }
    }
    if (mode == 3) {   // a regular k x k grid of quads with shared vertices in the plane z = 0: rows of equal centroid coordinates, a degenerate axis
        P.clear(); idx.clear();
        const size_t k = std::max<size_t>(1, (size_t)std::sqrt((double)n / 2.0));
        for (size_t j = 0; j <= k; j++) for (size_t i = 0; i <= k; i++) { P.push_back(-1.0f + 2.0f * (float)i / (float)k); P.push_back(-1.0f + 2.0f * (float)j / (float)k); P.push_back(0.0f); }
        for (size_t j = 0; j < k; j++) for (size_t i = 0; i < k; i++) {
            const uint32_t a = (uint32_t)(j * (k + 1) + i), b = a + 1, c = a + (uint32_t)k + 1, d = c + 1;
            const uint32_t q[6] = {a, b, d, a, d, c};
            idx.insert(idx.end(), q, q + 6);
        }
        n = idx.size() / 3;
    }
    phost::BuildInput in{P.data(), idx.data(), n, nullptr, nullptr};
    phost::BuildOutput want;
    // forest != 0: the triangles become a scene with object instances — the first quarter stays in the scene's own list, the rest is cut into `forest` object definitions, each
    // instanced once or twice under a random affine map (one object of a single triangle among them when there is room): the passes then build all the trees at once
    std::vector<uint32_t> tri0, tri1, inst_object, top_items; std::vector<float> inst_i2w;
    phost::ForestLayout layout; std::vector<phost::ForestTreeOut> want_trees;
    phost::InstancedScene isc{};
    if (forest) {
        const size_t n_top = n / 4;
        for (size_t t = 0; t < n_top; t++) top_items.push_back((uint32_t)t);
        size_t at = n_top;
        for (int k = 0; k < forest && at < n; k++) {
            size_t len = (k == 1) ? 1 : std::max<size_t>(1, (n - n_top) / (size_t)forest);
            if (k == forest - 1 || at + len > n) len = n - at;
            tri0.push_back((uint32_t)at); tri1.push_back((uint32_t)(at + len)); at += len;
        }
        for (size_t ob = 0; ob < tri0.size(); ob++)
            for (int rep = 0; rep < 1 + (int)(ob % 2); rep++) {
                float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
                for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) m[4 * r + q] = (r == q ? 0.5f : 0.0f) + 0.3f * U(rng); m[4 * r + 3] = U(rng); }
                inst_object.push_back((uint32_t)ob); inst_i2w.insert(inst_i2w.end(), m, m + 16);
                top_items.insert(top_items.begin() + (long)(rng() % (top_items.size() + 1)), PH_ITEM_INST | (uint32_t)(inst_object.size() - 1));
            }
        isc = phost::InstancedScene{tri0.data(), tri1.data(), tri0.size(), inst_object.data(), inst_i2w.data(), inst_object.size(), top_items.data(), top_items.size()};
        phost::forest_layout(isc, layout);
        if (phost::build_forest_host(in, isc, layout, 0, max_prims, want, want_trees) != 0) { std::printf("host forest build failed\n"); return 1; }
    } else if (phost::build_bvh(in, 0, max_prims, 1, want) != 0) { std::printf("host build failed\n"); return 1; }

    const uint32_t N = (uint32_t)(forest ? layout.items.size() : n);
    const uint32_t n_trees = forest ? (uint32_t)layout.tree_start.size() - 1u : 1u;
    const uint32_t one_tree[2] = {0u, N};
    std::vector<uint32_t> root_words(12 * (size_t)n_trees); std::vector<float> ibounds(6 * inst_object.size());
    std::vector<Elem> e_lo(N), e_hi(N); std::vector<uint32_t> seg(N), scan(N + 1), lf(N), lb(N), leaf_last(N, 0u), counters(16, 0u); std::vector<uint8_t> bkt(N);
    std::vector<SNode> nodes(2 * (size_t)N + 2); std::vector<uint32_t> scratch; std::vector<Node64> out_nodes(N); std::vector<TriRec> out_tris(N);
    Ctx c{};
    c.n = N; c.max_prims = (uint32_t)(max_prims & 0xff); c.P = P.data(); c.idx = idx.data(); c.tri_flags = nullptr; c.tri_mesh = nullptr;
    c.e_lo = e_lo.data(); c.e_hi = e_hi.data(); c.seg = seg.data(); c.bkt = bkt.data(); c.nodes = nodes.data(); c.counters = counters.data();
    c.scan = scan.data(); c.lf = lf.data(); c.lb = lb.data(); c.leaf_last = leaf_last.data(); c.out_nodes = out_nodes.data(); c.out_tris = out_tris.data();
    c.items = forest ? layout.items.data() : nullptr; c.inst_bounds = ibounds.data(); c.tree_start = forest ? layout.tree_start.data() : one_tree; c.n_trees = n_trees; c.root_words = root_words.data();
    for (uint32_t t = 0; t < n_trees; t++) init_bounds_words(c.root_words + 12 * t);
    const uint32_t top_end = forest ? layout.tree_start[1] : 0u;
    for (uint32_t i = top_end; i < N; i++) { const uint32_t t = tree_of(c, i); init_elem(c, i, t, c.root_words + 12 * t); }   // the objects first: the instances' bounds come from their roots
    for (size_t k = 0; k < inst_object.size(); k++) {
        const uint32_t t = layout.inst_tree[k];
        float lo[3], hi[3];
        for (int q = 0; q < 3; q++) { lo[q] = ord2f(root_words[12 * t + q]); hi[q] = ord2f(root_words[12 * t + 3 + q]); }
        phost::transform_bounds(&inst_i2w[16 * k], lo, hi, &ibounds[6 * k]);
    }
    for (uint32_t i = 0; i < top_end; i++) init_elem(c, i, 0u, c.root_words);
    for (uint32_t t = 0; t < n_trees; t++) make_root(c, t);
    std::vector<std::pair<uint32_t, uint32_t>> levels;
    uint32_t lb0 = 0, le0 = n_trees; int depth = 0;
    while (lb0 < le0) {
        levels.push_back({lb0, le0});
        c.counters[1] = 0;
        for (uint32_t v = lb0; v < le0; v++) decide_node(c, v);
        const uint32_t slots = c.counters[1];
        if (slots) {
            scratch.resize((size_t)slots * PHS_SLOT);
            for (size_t w = 0; w < scratch.size(); w++) scratch[w] = scratch_init_word((uint32_t)(w % PHS_SLOT));
            c.scratch = scratch.data();
            for (uint32_t i = 0; i < N; i++) bucket_elem(c, i, nullptr);
            for (uint32_t v = lb0; v < le0; v++) eval_node(c, v);
            uint32_t run = 0;
            for (uint32_t i = 0; i < N; i++) { scan[i] = run; run += flag_of(c, i); }
            scan[N] = run;
            for (uint32_t i = 0; i < N; i++) list_elem(c, i);
            for (uint32_t i = 0; i < N; i++) swap_pos(c, i);
        }
        for (uint32_t i = 0; i < N; i++) relabel_pos(c, i);
        if (c.counters[0] > le0) depth = (int)levels.size();
        lb0 = le0; le0 = c.counters[0];
    }
    for (size_t L = levels.size(); L-- > 0;) for (uint32_t v = levels[L].first; v < levels[L].second; v++) size_node(c, v);
    uint32_t interior = 0;
    for (uint32_t t = 0; t < n_trees; t++) { root_numbers(c, t, interior); interior += nodes[t].size; }
    for (size_t L = 0; L < levels.size(); L++) for (uint32_t v = levels[L].first; v < levels[L].second; v++) number_node(c, v);
    for (uint32_t i = 0; i < N; i++) emit_tri(c, i);

    int bad = 0;
    if (interior != want.interior_nodes || c.counters[2] != want.leaf_nodes || c.counters[3] != want.max_leaf_prims || depth != want.max_depth) {
        std::printf("n=%zu seed=%u: counts differ: interior %u/%zu leaves %u/%zu maxleaf %u/%zu depth %d/%d\n", n, seed, interior, want.interior_nodes, c.counters[2], want.leaf_nodes,
                    c.counters[3], want.max_leaf_prims, depth, want.max_depth);
        bad++;
    }
    for (uint32_t i = 0; i < N && bad < 5; i++)
        if (out_tris[i].prim != want.tris[i].prim || out_tris[i].flags != want.tris[i].flags) { std::printf("n=%zu seed=%u: tri %u: %u/%u flags %u/%u\n", n, seed, i, out_tris[i].prim, want.tris[i].prim, out_tris[i].flags, want.tris[i].flags); bad++; }
    for (uint32_t v = 0; v < interior && v < want.nodes.size() && bad < 5; v++) {
        const Node64 &a = out_nodes[v], &b = want.nodes[v];
        const float* fa = a.x0; const float* fb = b.x0;
        bool same = a.c0 == b.c0 && a.c1 == b.c1 && a.axis == b.axis;
        for (int k = 0; k < 12; k++) same = same && fa[k] == fb[k];
        if (!same) { std::printf("n=%zu seed=%u: node %u differs (c0 %x/%x c1 %x/%x axis %u/%u)\n", n, seed, v, a.c0, b.c0, a.c1, b.c1, a.axis, b.axis); bad++; }
    }
    for (int k = 0; k < 3; k++) if (nodes[0].lo[k] != want.root_lo[k] || nodes[0].hi[k] != want.root_hi[k]) { std::printf("root bound differs\n"); bad++; }
    uint32_t base = 0;
    for (uint32_t t = 0; forest && t < n_trees; t++) {   // every tree's root as the instance records will name it
        const uint32_t ref = nodes[t].size ? base : (PH_LEAF_BIT | layout.tree_start[t]);
        base += nodes[t].size;
        bool same = ref == want_trees[t].root_ref;
        for (int k = 0; k < 3; k++) same = same && nodes[t].lo[k] == want_trees[t].lo[k] && nodes[t].hi[k] == want_trees[t].hi[k];
        if (!same) { std::printf("n=%zu seed=%u forest=%d: tree %u: root %x/%x or its bound differs\n", n, seed, forest, t, ref, want_trees[t].root_ref); bad++; }
    }
    return bad;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "quick";   // the subset tests/test_sah_steps_cpu.py runs
    int bad = 0, cases = 0;
    const std::vector<size_t> sizes = quick ? std::vector<size_t>{1, 2, 3, 5, 17, 300, 5000} : std::vector<size_t>{1, 2, 3, 4, 5, 7, 17, 64, 300, 1000, 5000, 40000};
    for (size_t n : sizes)
        for (int mode = 0; mode < 4; mode++)
            for (int mp : {1, 4, 8, 255})
                for (unsigned seed = 1; seed <= (quick ? 1u : (n <= 300 ? 6u : 2u)); seed++) { bad += run(n, seed * 7919u + (unsigned)n, mp, mode); cases++; }
    for (size_t n : (quick ? std::vector<size_t>{9, 300, 5000} : std::vector<size_t>{5, 9, 40, 300, 5000, 40000}))
        for (int forest : {1, 3, 17})
            for (int mp : {1, 4})
                for (unsigned seed = 1; seed <= (quick ? 1u : 3u); seed++) { bad += run(n, seed * 104729u + (unsigned)n, mp, 0, forest); cases++; }
    std::printf("%d cases, %d differences\n", cases, bad);
    return bad ? 1 : 0;
}
