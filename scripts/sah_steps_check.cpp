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

static int run(size_t n, unsigned seed, int max_prims, int mode) {
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
    if (phost::build_bvh(in, 0, max_prims, 1, want) != 0) { std::printf("host build failed\n"); return 1; }

    const uint32_t N = (uint32_t)n;
    std::vector<Elem> e_lo(N), e_hi(N); std::vector<uint32_t> seg(N), scan(N + 1), lf(N), lb(N), leaf_last(N, 0u), counters(16, 0u); std::vector<uint8_t> bkt(N);
    std::vector<SNode> nodes(2 * (size_t)N + 2); std::vector<uint32_t> scratch; std::vector<Node64> out_nodes(N); std::vector<TriRec> out_tris(N);
    Ctx c{};
    c.n = N; c.max_prims = (uint32_t)(max_prims & 0xff); c.P = P.data(); c.idx = idx.data(); c.tri_flags = nullptr; c.tri_mesh = nullptr;
    c.e_lo = e_lo.data(); c.e_hi = e_hi.data(); c.seg = seg.data(); c.bkt = bkt.data(); c.nodes = nodes.data(); c.counters = counters.data();
    c.scan = scan.data(); c.lf = lf.data(); c.lb = lb.data(); c.leaf_last = leaf_last.data(); c.out_nodes = out_nodes.data(); c.out_tris = out_tris.data();
    init_bounds_words(c.counters + 4);
    for (uint32_t i = 0; i < N; i++) init_elem(c, i, c.counters + 4);
    make_root(c);
    std::vector<std::pair<uint32_t, uint32_t>> levels;
    uint32_t lb0 = 0, le0 = 1; int depth = 0;
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
    for (size_t L = 0; L < levels.size(); L++) for (uint32_t v = levels[L].first; v < levels[L].second; v++) number_node(c, v);
    for (uint32_t i = 0; i < N; i++) emit_tri(c, i);

    int bad = 0;
    const uint32_t interior = nodes[0].size;
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
    std::printf("%d cases, %d differences\n", cases, bad);
    return bad ? 1 : 0;
}
