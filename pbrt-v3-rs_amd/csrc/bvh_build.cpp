// See bvh_build.h.  Host C++ only (no device code).
#include "bvh_build.h"
#include "guard.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

namespace phost {
namespace {

inline float fmn(float a, float b) { return a < b ? a : b; }  // core/src/pbrt/common.rs:81-92 (`<`-based)
inline float fmx(float a, float b) { return a > b ? a : b; }

struct Box {
    float lo[3], hi[3];
    void reset() {  // Bounds3f::EMPTY (bounds3.rs:23-30)
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::max(); hi[k] = std::numeric_limits<float>::lowest(); }
    }
    void grow(const Box& o) { for (int k = 0; k < 3; k++) { lo[k] = fmn(lo[k], o.lo[k]); hi[k] = fmx(hi[k], o.hi[k]); } }
    void grow_pt(const float* p) { for (int k = 0; k < 3; k++) { lo[k] = fmn(lo[k], p[k]); hi[k] = fmx(hi[k], p[k]); } }
    float area() const {  // bounds3.rs:95-107
        if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return 0.0f;
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        float h = dx * dy + dx * dz + dy * dz;
        return h + h;
    }
    int widest() const {  // bounds3.rs:122-134
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        return dy > dz ? 1 : 2;
    }
};

struct Prim { uint32_t id; Box b; float c[3]; };  // BVHPrimitiveInfo (bvh/common.rs:62-91)

struct BNode {
    Box b;
    BNode* kid[2];
    uint32_t first, count;  // leaf: range in the (final) prim array
    int axis;
};

constexpr int kBuckets = 12;

struct Builder {
    std::vector<Prim> prims;
    std::vector<BNode> pool;
    std::atomic<size_t> pool_next{0};
    std::atomic<int> spare_threads{0};
    int split_method = 0, max_prims = 4;

    // Nodes come from per-thread chunks of the pool: one shared atomic per 1024 nodes instead of one per node.
    static constexpr size_t kArenaChunk = 1024;
    static uint64_t next_serial() { static std::atomic<uint64_t> g{1}; return g.fetch_add(1); }
    const uint64_t serial = next_serial();
    BNode* alloc() {
        thread_local BNode* cur = nullptr; thread_local BNode* lim = nullptr; thread_local uint64_t owner = 0;
        if (owner != serial || cur == lim) {  // `serial` is unique per Builder: a later Builder at the same address must not reuse the chunk
            const size_t at = pool_next.fetch_add(kArenaChunk);
            cur = pool.data() + at; lim = cur + kArenaChunk; owner = serial;
        }
        return cur++;
    }

    // Helper threads for the data-parallel passes over one large range (taken from the same budget as the forked subtrees).
    static constexpr size_t kParallelRange = 1u << 18;
    int grab_helpers() {
        int got = 0;
        while (got < 7) { if (spare_threads.fetch_sub(1) > 0) got++; else { spare_threads.fetch_add(1); break; } }
        return got;
    }
    void release_helpers(int n) { if (n > 0) spare_threads.fetch_add(n); }
    template <class F> static void parallel_chunks(size_t start, size_t end, int helpers, F f) {
        const size_t n = end - start, parts = (size_t)helpers + 1, step = (n + parts - 1) / parts;
        ThreadGroup th;
        for (int t = 1; t <= helpers; t++) {
            const size_t a = std::min(end, start + (size_t)t * step), b = std::min(end, a + step);
            th.run([=]() { f(t, a, b); });
        }
        f(0, start, std::min(end, start + step));
        th.join();
    }

    static uint32_t bucket_of(const Box& cb, const float* c, int dim) {  // sah.rs:309-313 (saturating `as usize`)
        float o = c[dim] - cb.lo[dim];
        if (cb.hi[dim] > cb.lo[dim]) o /= cb.hi[dim] - cb.lo[dim];
        float f = (float)kBuckets * o;
        uint32_t b = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f);
        if (b == (uint32_t)kBuckets) b = kBuckets - 1;
        return b;
    }

    // itertools::partition — the exact element order matters for leaf order
    template <class Pred> size_t partition_like_itertools(size_t start, size_t end, Pred pred) {
        size_t split = 0, front = start, back = end;
        while (front != back) {
            size_t f = front++;
            if (!pred(prims[f])) {
                bool swapped = false;
                while (front != back) {
                    size_t b = --back;
                    if (pred(prims[b])) { std::swap(prims[f], prims[b]); swapped = true; break; }
                }
                if (!swapped) return split;
            }
            split++;
        }
        return split;
    }

    BNode* make_leaf(BNode* n, const Box& bounds, size_t start, size_t end) {
        n->b = bounds; n->kid[0] = n->kid[1] = nullptr; n->first = (uint32_t)start; n->count = (uint32_t)(end - start); n->axis = 0;
        return n;
    }

    BNode* build(size_t start, size_t end) {
        BNode* node = alloc();
        Box bounds, cb; bounds.reset(); cb.reset();
        const size_t n = end - start;
        // one pass for the node bound and the centroid bound; min/max are exact, so any grouping of the unions gives the reference's boxes.
        // Large ranges (the top of the tree, where nothing else runs in parallel yet) are reduced by several threads.
        const int helpers = n >= kParallelRange ? grab_helpers() : 0;
        if (helpers > 0) {
            std::vector<Box> pb(helpers + 1), pc(helpers + 1);
            parallel_chunks(start, end, helpers, [&](int t, size_t a, size_t b) {
                Box bb, cc; bb.reset(); cc.reset();
                for (size_t i = a; i < b; i++) { bb.grow(prims[i].b); cc.grow_pt(prims[i].c); }
                pb[t] = bb; pc[t] = cc;
            });
            for (int t = 0; t <= helpers; t++) { bounds.grow(pb[t]); cb.grow(pc[t]); }
        } else {
            for (size_t i = start; i < end; i++) { bounds.grow(prims[i].b); cb.grow_pt(prims[i].c); }
        }
        if (n == 1) { release_helpers(helpers); return make_leaf(node, bounds, start, end); }
        const int dim = cb.widest();
        if (cb.hi[dim] == cb.lo[dim]) { release_helpers(helpers); return make_leaf(node, bounds, start, end); }  // sah.rs:61-63

        size_t mid;
        if (split_method == 3 || n <= 2) {
            // split_equal_counts (sah.rs:240-254).  For n == 2 every correct selection yields [smaller, larger].
            mid = (start + end) / 2;
            if (n == 2) { if (prims[start + 1].c[dim] < prims[start].c[dim]) std::swap(prims[start], prims[start + 1]); }
            else std::nth_element(prims.begin() + start, prims.begin() + mid, prims.begin() + end,
                                  [dim](const Prim& a, const Prim& b) { return a.c[dim] < b.c[dim]; });
        } else {
            // split_sah (sah.rs:293-367)
            size_t cnt[kBuckets]; Box bb[kBuckets];
            for (int i = 0; i < kBuckets; i++) { cnt[i] = 0; bb[i].reset(); }
            if (helpers > 0) {
                struct Part { size_t cnt[kBuckets]; Box bb[kBuckets]; };
                std::vector<Part> parts(helpers + 1);
                parallel_chunks(start, end, helpers, [&](int t, size_t a, size_t b2) {
                    Part& q = parts[t];
                    for (int i = 0; i < kBuckets; i++) { q.cnt[i] = 0; q.bb[i].reset(); }
                    for (size_t i = a; i < b2; i++) { uint32_t b = bucket_of(cb, prims[i].c, dim); q.cnt[b]++; q.bb[b].grow(prims[i].b); }
                });
                for (int t = 0; t <= helpers; t++) for (int i = 0; i < kBuckets; i++) { cnt[i] += parts[t].cnt[i]; bb[i].grow(parts[t].bb[i]); }
            } else
            for (size_t i = start; i < end; i++) { uint32_t b = bucket_of(cb, prims[i].c, dim); cnt[b]++; bb[b].grow(prims[i].b); }
            // suffix unions once; prefix grown on the fly (min/max are exact, so the grouping does not matter)
            Box suf[kBuckets]; size_t sufc[kBuckets];
            suf[kBuckets - 1] = bb[kBuckets - 1]; sufc[kBuckets - 1] = cnt[kBuckets - 1];
            for (int i = kBuckets - 2; i >= 0; i--) { suf[i] = suf[i + 1]; suf[i].grow(bb[i]); sufc[i] = sufc[i + 1] + cnt[i]; }
            Box pre; pre.reset(); size_t prec = 0;
            const float inv_denominator_area = bounds.area();
            float best = 0.0f; int best_b = 0;
            for (int i = 0; i < kBuckets - 1; i++) {
                pre.grow(bb[i]); prec += cnt[i];
                float cost = 1.0f + ((float)prec * pre.area() + (float)sufc[i + 1] * suf[i + 1].area()) / inv_denominator_area;
                if (i == 0 || cost < best) { best = cost; best_b = i; }
            }
            const float leaf_cost = (float)n;
            if (n > (size_t)max_prims || best < leaf_cost) {
                mid = start + partition_like_itertools(start, end, [&](const Prim& p) { return bucket_of(cb, p.c, dim) <= (uint32_t)best_b; });
            } else { release_helpers(helpers); return make_leaf(node, bounds, start, end); }
        }
        release_helpers(helpers);
        if (mid == start || mid == end) return make_leaf(node, bounds, start, end);  // reference: assert_ne! panic (sah.rs:37)

        node->axis = dim; node->first = 0; node->count = 0;
        // fork the larger subtrees onto spare threads
        if (n > 32768 && spare_threads.fetch_sub(1) > 0) {
            BNode* left = nullptr;
            ThreadGroup t;
            t.run([&]() { left = build(start, mid); });
            node->kid[1] = build(mid, end);
            t.join();
            node->kid[0] = left;
            spare_threads.fetch_add(1);
        } else {
            if (n > 32768) spare_threads.fetch_add(1);
            node->kid[0] = build(start, mid);
            node->kid[1] = build(mid, end);
        }
        node->b = node->kid[0]->b; node->b.grow(node->kid[1]->b);  // bvh/common.rs:152-160
        return node;
    }

    // ================= HLBVH (accelerators/src/bvh/hlbvh.rs:33-449, morton.rs) =========================================
    // Same tree as the reference builds, including its Morton quirk: encode_morton_3 interleaves the low bits of the IEEE
    // bit pattern of the scaled centroid offset (morton.rs:33-39 `float_to_bits`), so the order is not spatial and the
    // tree is valid but slow to traverse.  Treelets are independent and built by a thread pool; the reference hands out
    // ordered_prims offsets in completion order, here leaves are laid out depth first afterwards (nothing a ray can
    // observe depends on it).
    struct MP { uint32_t id, code; };
    std::vector<MP> mp;
    std::atomic<bool> panic{false};

    static uint32_t left_shift_3(uint32_t x) {  // morton.rs:101-118
        uint32_t v = (x == (1u << 10)) ? x - 1 : x;
        v = (v | (v << 16)) & 0x030000FFu;
        v = (v | (v << 8)) & 0x0300F00Fu;
        v = (v | (v << 4)) & 0x030C30C3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    }
    static uint32_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

    BNode* emit_lbvh(size_t first, size_t n, int bit) {  // hlbvh.rs:199-294; leaves refer to [first, first+n) of `mp`
        for (;;) {
            if (bit == -1 || n < (size_t)max_prims) {
                BNode* leaf = alloc();
                leaf->b.reset();
                for (size_t i = 0; i < n; i++) leaf->b.grow(prims[mp[first + i].id].b);
                leaf->kid[0] = leaf->kid[1] = nullptr; leaf->first = (uint32_t)first; leaf->count = (uint32_t)n; leaf->axis = 0;
                return leaf;
            }
            const uint32_t mask = 1u << bit;
            if ((mp[first].code & mask) != (mp[first + n - 1].code & mask)) break;
            bit--;  // no split on this bit
        }
        const uint32_t mask = 1u << bit;
        size_t lo = 0, hi = n - 1;
        while (lo + 1 != hi) {
            const size_t mid = (lo + hi) / 2;
            if ((mp[first + lo].code & mask) == (mp[first + mid].code & mask)) lo = mid; else hi = mid;
        }
        BNode* node = alloc();
        node->kid[0] = emit_lbvh(first, hi, bit - 1);
        node->kid[1] = emit_lbvh(first + hi, n - hi, bit - 1);
        node->axis = bit % 3; node->first = 0; node->count = 0;
        node->b = node->kid[0]->b; node->b.grow(node->kid[1]->b);
        return node;
    }
    static uint32_t hl_bucket(float c, float lo, float hi) {  // hlbvh.rs:349-355 (saturating `as usize`)
        const float f = (float)kBuckets * ((c - lo) / (hi - lo));
        uint32_t b = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f);
        if (b == (uint32_t)kBuckets) b = kBuckets - 1;
        return b;
    }
    BNode* upper_sah(std::vector<BNode*>& roots, size_t start, size_t end) {  // hlbvh.rs:296-432
        if (end - start == 1) return roots[start];
        Box bounds, cb; bounds.reset(); cb.reset();
        for (size_t i = start; i < end; i++) bounds.grow(roots[i]->b);
        for (size_t i = start; i < end; i++) {
            float c[3];
            for (int k = 0; k < 3; k++) c[k] = (roots[i]->b.lo[k] + roots[i]->b.hi[k]) * 0.5f;
            cb.grow_pt(c);
        }
        const int dim = cb.widest();
        if (cb.hi[dim] == cb.lo[dim]) { panic = true; return roots[start]; }  // assert_ne! (:338)
        size_t cnt[kBuckets]; Box bb[kBuckets];
        for (int i = 0; i < kBuckets; i++) { cnt[i] = 0; bb[i].reset(); }
        auto bucket = [&](const BNode* r) { return hl_bucket((r->b.lo[dim] + r->b.hi[dim]) * 0.5f, cb.lo[dim], cb.hi[dim]); };
        for (size_t i = start; i < end; i++) {
            const uint32_t b = bucket(roots[i]);
            if (b >= (uint32_t)kBuckets) { panic = true; return roots[start]; }
            cnt[b]++; bb[b].grow(roots[i]->b);
        }
        float best = 0.0f; int best_b = 0;
        for (int i = 0; i < kBuckets - 1; i++) {
            Box b0, b1; b0.reset(); b1.reset(); size_t c0 = 0, c1 = 0;
            for (int j = 0; j <= i; j++) { b0.grow(bb[j]); c0 += cnt[j]; }
            for (int j = i + 1; j < kBuckets; j++) { b1.grow(bb[j]); c1 += cnt[j]; }
            const float cost = 0.125f + ((float)c0 * b0.area() + (float)c1 * b1.area()) / bounds.area();
            if (i == 0 || cost < best) { best = cost; best_b = i; }
        }
        size_t split = 0, front = start, back = end;  // itertools::partition
        while (front != back) {
            const size_t f = front++;
            if (!(bucket(roots[f]) <= (uint32_t)best_b)) {
                bool swapped = false;
                while (front != back) {
                    const size_t b = --back;
                    if (bucket(roots[b]) <= (uint32_t)best_b) { std::swap(roots[f], roots[b]); swapped = true; break; }
                }
                if (!swapped) break;
            }
            split++;
        }
        const size_t mid = start + split;
        if (!(mid > start) || !(mid < end)) { panic = true; return roots[start]; }  // assert! (:418-419)
        BNode* node = alloc();
        node->kid[0] = upper_sah(roots, start, mid);
        node->kid[1] = upper_sah(roots, mid, end);
        node->axis = dim; node->first = 0; node->count = 0;
        node->b = node->kid[0]->b; node->b.grow(node->kid[1]->b);
        return node;
    }
    BNode* build_hlbvh(int n_threads) {
        const size_t n = prims.size();
        Box bounds; bounds.reset();
        for (const Prim& p : prims) bounds.grow(p.b);
        mp.resize(n);
        for (size_t i = 0; i < n; i++) {  // compute_morton_primitives (:97-135): Bounds3::offset * 1024, then the BITS of each float
            uint32_t c[3];
            for (int k = 0; k < 3; k++) {
                float o = prims[i].c[k] - bounds.lo[k];
                if (bounds.hi[k] > bounds.lo[k]) o /= bounds.hi[k] - bounds.lo[k];
                c[k] = left_shift_3(bits_of(o * 1024.0f));
            }
            mp[i].id = (uint32_t)i; mp[i].code = (c[2] << 2) | (c[1] << 1) | c[0];
        }
        {  // radix_sort (morton.rs:50-98): five stable 6-bit passes
            std::vector<MP> tmp(n);
            for (int pass = 0; pass < 5; pass++) {
                const int low = pass * 6;
                std::vector<MP>& in = (pass & 1) ? tmp : mp;
                std::vector<MP>& out = (pass & 1) ? mp : tmp;
                size_t count[64] = {0}, at[64];
                for (const MP& m : in) count[(m.code >> low) & 63]++;
                at[0] = 0;
                for (int i = 1; i < 64; i++) at[i] = at[i - 1] + count[i - 1];
                for (const MP& m : in) out[at[(m.code >> low) & 63]++] = m;
            }
            mp.swap(tmp);
        }
        struct Span { size_t first, n; };
        std::vector<Span> spans;
        const uint32_t MASK = 0x3FFC0000u;
        for (size_t start = 0, end = 1; end <= n; end++)
            if (end == n || ((mp[start].code & MASK) != (mp[end].code & MASK))) { spans.push_back({start, end - start}); start = end; }
        std::vector<BNode*> roots(spans.size());
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < spans.size();) roots[i] = emit_lbvh(spans[i].first, spans[i].n, 29 - 12); };
        ThreadGroup pool_threads;
        for (int t = 1; t < n_threads && (size_t)t < spans.size(); t++) pool_threads.run(work);
        work();
        pool_threads.join();
        BNode* root = upper_sah(roots, 0, roots.size());
        // lay the leaves out depth first: prims[] becomes the final primitive order, leaf ranges point into it
        std::vector<Prim> ordered; ordered.reserve(n);
        std::vector<BNode*> stack{root};
        while (!stack.empty()) {
            BNode* nd = stack.back(); stack.pop_back();
            if (!nd->kid[0]) {
                const uint32_t first = (uint32_t)ordered.size();
                for (uint32_t i = 0; i < nd->count; i++) ordered.push_back(prims[mp[nd->first + i].id]);
                nd->first = first;
            } else { stack.push_back(nd->kid[1]); stack.push_back(nd->kid[0]); }
        }
        prims.swap(ordered);
        return root;
    }
};

}  // namespace

int build_upper_sah(const float* root_bounds, size_t n_roots, std::vector<UpperNode>& out_nodes, int& out_root) {
    out_nodes.clear(); out_root = -1;
    if (n_roots == 0) return 0;
    Builder B;
    B.pool.resize(2 * n_roots + 2 * Builder::kArenaChunk);
    std::vector<BNode> leaves(n_roots);
    std::vector<BNode*> roots(n_roots);
    for (size_t t = 0; t < n_roots; t++) {
        BNode& l = leaves[t];
        for (int k = 0; k < 3; k++) { l.b.lo[k] = root_bounds[6 * t + k]; l.b.hi[k] = root_bounds[6 * t + 3 + k]; }
        l.kid[0] = l.kid[1] = nullptr; l.first = (uint32_t)t; l.count = 1; l.axis = 0;
        roots[t] = &l;
    }
    BNode* root = B.upper_sah(roots, 0, n_roots);
    if (B.panic) return -2;
    // pre-order flattening; a node that is one of `leaves` stands for its treelet
    auto is_treelet = [&](const BNode* nd) { return nd >= leaves.data() && nd < leaves.data() + n_roots; };
    if (is_treelet(root)) { out_root = -1 - (int)(root - leaves.data()); return 0; }
    struct Item { const BNode* n; int self; };
    std::vector<Item> stack;
    out_nodes.emplace_back();
    out_root = 0;
    stack.push_back({root, 0});
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        int kid[2];
        for (int c = 0; c < 2; c++) {
            const BNode* k = it.n->kid[c];
            if (is_treelet(k)) kid[c] = -1 - (int)(k - leaves.data());
            else { kid[c] = (int)out_nodes.size(); out_nodes.emplace_back(); }
        }
        UpperNode& u = out_nodes[(size_t)it.self];
        for (int k = 0; k < 3; k++) { u.lo[k] = it.n->b.lo[k]; u.hi[k] = it.n->b.hi[k]; }
        u.kid[0] = kid[0]; u.kid[1] = kid[1]; u.axis = it.n->axis;
        if (kid[1] >= 0) stack.push_back({it.n->kid[1], kid[1]});
        if (kid[0] >= 0) stack.push_back({it.n->kid[0], kid[0]});
    }
    return 0;
}

int build_bvh(const BuildInput& in, int split_method, int max_prims_in_node, int n_threads, BuildOutput& out) {
    if (split_method != 0 && split_method != 1 && split_method != 3) return -1;
    out = BuildOutput();
    const size_t n = in.items ? in.n_items : in.n_tris;
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFu) return -1;
    auto t0 = std::chrono::steady_clock::now();
    Builder B;
    B.split_method = split_method;
    B.max_prims = max_prims_in_node & 0xff;  // bvh/mod.rs:357 `as u8`
    B.prims.resize(n);
    for (size_t i = 0; i < n; i++) {  // Triangle::world_bound (triangle.rs:427-431) + BVHPrimitiveInfo::new
        Prim& p = B.prims[i];
        p.id = in.items ? in.items[i] : (uint32_t)i;
        if (p.id & PH_ITEM_INST) {  // TransformedPrimitive::world_bound, computed by the caller
            const float* bb = in.inst_bounds + 6 * (size_t)(p.id & ~PH_ITEM_INST);
            for (int k = 0; k < 3; k++) { p.b.lo[k] = bb[k]; p.b.hi[k] = bb[3 + k]; }
        } else {
            const size_t t = p.id;
            const float* a = in.P + 3 * (size_t)in.idx[3 * t];
            p.b.lo[0] = p.b.hi[0] = a[0]; p.b.lo[1] = p.b.hi[1] = a[1]; p.b.lo[2] = p.b.hi[2] = a[2];
            p.b.grow_pt(in.P + 3 * (size_t)in.idx[3 * t + 1]);
            p.b.grow_pt(in.P + 3 * (size_t)in.idx[3 * t + 2]);
        }
        for (int k = 0; k < 3; k++) p.c[k] = 0.5f * (p.b.lo[k] + p.b.hi[k]);
    }
    const bool prof = std::getenv("PBRT_HIP_BUILD_PROFILE") != nullptr;
    auto t_prims = std::chrono::steady_clock::now();
    B.pool.resize(2 * n + 4096 * Builder::kArenaChunk);  // room for the partly used chunks of every thread the build may start
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    B.spare_threads = n_threads > 1 ? n_threads - 1 : 0;
    BNode* root = split_method == 1 ? B.build_hlbvh(std::max(n_threads, 1)) : B.build(0, n);
    if (B.panic) return -2;  // one of the reference's assertions fired (hlbvh.rs:338/356/418)
    auto t_built = std::chrono::steady_clock::now();

    // ---- emit device layout: TriRecs in final prim-array order (= depth-first leaf order), Node64s in pre-order -------
    out.tris.resize(n);
    auto fill = [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
        const uint32_t id = B.prims[i].id;
        TriRec& t = out.tris[i];
        if (id & PH_ITEM_INST) { std::memset(&t, 0, sizeof t); t.prim = id & ~PH_ITEM_INST; t.flags = PH_TRI_INSTANCE; continue; }
        const float* p0 = in.P + 3 * (size_t)in.idx[3 * (size_t)id];
        const float* p1 = in.P + 3 * (size_t)in.idx[3 * (size_t)id + 1];
        const float* p2 = in.P + 3 * (size_t)in.idx[3 * (size_t)id + 2];
        std::memcpy(t.p0, p0, 12); std::memcpy(t.p1, p1, 12); std::memcpy(t.p2, p2, 12);
        t.prim = id; t.flags = in.tri_flags ? (in.tri_flags[id] & ~PH_TRI_LAST) : 0u; t.mesh = in.tri_mesh ? in.tri_mesh[id] : 0u;
    }
    };
    {  // the gather of vertex positions is a random walk over P: spread it over the threads
        const int nt = (n >= (1u << 16)) ? std::max(n_threads, 1) : 1;
        ThreadGroup th;
        const size_t step = (n + nt - 1) / nt;
        for (int t = 1; t < nt; t++) { const size_t a = std::min(n, (size_t)t * step), b = std::min(n, (size_t)(t + 1) * step); th.run([&fill, a, b]() { fill(a, b); }); }
        fill(0, std::min(n, step));
        th.join();
    }
    for (int k = 0; k < 3; k++) { out.root_lo[k] = root->b.lo[k]; out.root_hi[k] = root->b.hi[k]; }

    auto ref_of_leaf = [&](const BNode* l) {
        out.tris[l->first + l->count - 1].flags |= PH_TRI_LAST;
        out.leaf_nodes++;
        out.max_leaf_prims = std::max<size_t>(out.max_leaf_prims, l->count);
        return PH_LEAF_BIT | l->first;
    };
    if (!root->kid[0]) {
        out.root_ref = ref_of_leaf(root);
    } else {
        // iterative pre-order walk; an interior node's Node64 index is assigned when it is first reached
        struct Item { const BNode* n; uint32_t self; int depth; };
        std::vector<Item> stack;
        out.nodes.reserve(n);
        out.nodes.emplace_back();
        out.root_ref = 0;
        stack.push_back({root, 0u, 1});
        while (!stack.empty()) {
            Item it = stack.back(); stack.pop_back();
            out.interior_nodes++;
            out.max_depth = std::max(out.max_depth, it.depth);
            uint32_t refs[2];
            const BNode* kids[2] = {it.n->kid[0], it.n->kid[1]};
            // reserve indices so that child 0's subtree follows its parent contiguously (reference pre-order)
            for (int c = 0; c < 2; c++) {
                if (kids[c]->kid[0]) { refs[c] = (uint32_t)out.nodes.size(); out.nodes.emplace_back(); }
                else refs[c] = ref_of_leaf(kids[c]);
            }
            Node64& d = out.nodes[it.self];
            d.x0[0] = kids[0]->b.lo[0]; d.x0[1] = kids[0]->b.hi[0]; d.y0[0] = kids[0]->b.lo[1]; d.y0[1] = kids[0]->b.hi[1];
            d.z0[0] = kids[0]->b.lo[2]; d.z0[1] = kids[0]->b.hi[2];
            d.x1[0] = kids[1]->b.lo[0]; d.x1[1] = kids[1]->b.hi[0]; d.y1[0] = kids[1]->b.lo[1]; d.y1[1] = kids[1]->b.hi[1];
            d.z1[0] = kids[1]->b.lo[2]; d.z1[1] = kids[1]->b.hi[2];
            d.c0 = refs[0]; d.c1 = refs[1]; d.axis = (uint32_t)it.n->axis; d.pad = 0;
            if (kids[1]->kid[0]) stack.push_back({kids[1], refs[1], it.depth + 1});
            if (kids[0]->kid[0]) stack.push_back({kids[0], refs[0], it.depth + 1});
        }
    }
    if (prof) {
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "build_bvh n=%zu: prims %.3f s, tree %.3f s, emit %.3f s\n", n, std::chrono::duration<double>(t_prims - t0).count(),
                     std::chrono::duration<double>(t_built - t_prims).count(), std::chrono::duration<double>(now - t_built).count());
    }
    out.total_nodes = out.interior_nodes + out.leaf_nodes;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

void forest_layout(const InstancedScene& sc, ForestLayout& out) {
    out.items.assign(sc.top_items, sc.top_items + sc.n_top);
    out.tree_start.assign({0u, (uint32_t)sc.n_top});
    out.inst_tree.assign(sc.n_inst, 0u);
    std::vector<uint32_t> tree_of_object(sc.n_objects, 0u);
    for (size_t k = 0; k < sc.n_inst; k++) {
        const uint32_t ob = sc.inst_object[k];
        if (!tree_of_object[ob]) {   // make_accelerator at the object's first ObjectInstance (lib.rs:953-971)
            tree_of_object[ob] = (uint32_t)out.tree_start.size() - 1u;
            for (uint32_t t = sc.obj_tri0[ob]; t < sc.obj_tri1[ob]; t++) out.items.push_back(t);
            out.tree_start.push_back((uint32_t)out.items.size());
        }
        out.inst_tree[k] = tree_of_object[ob];
    }
}

void transform_bounds(const float* m, const float* lo, const float* hi, float* ob) {
    auto xf = [&](float x, float y, float z, float* o) {
        const float xp = m[0] * x + m[1] * y + m[2] * z + m[3], yp = m[4] * x + m[5] * y + m[6] * z + m[7];
        const float zp = m[8] * x + m[9] * y + m[10] * z + m[11], wp = m[12] * x + m[13] * y + m[14] * z + m[15];
        if (wp == 1.0f) { o[0] = xp; o[1] = yp; o[2] = zp; } else { const float inv = 1.0f / wp; o[0] = inv * xp; o[1] = inv * yp; o[2] = inv * zp; }
    };
    const float corners[8][3] = {{lo[0], lo[1], lo[2]}, {hi[0], lo[1], lo[2]}, {lo[0], hi[1], lo[2]}, {lo[0], lo[1], hi[2]},
                                 {lo[0], hi[1], hi[2]}, {hi[0], hi[1], lo[2]}, {hi[0], lo[1], hi[2]}, {hi[0], hi[1], hi[2]}};
    for (int c = 0; c < 8; c++) {
        float q[3]; xf(corners[c][0], corners[c][1], corners[c][2], q);
        for (int a = 0; a < 3; a++) {
            if (c == 0) { ob[a] = ob[3 + a] = q[a]; }
            else { ob[a] = ob[a] < q[a] ? ob[a] : q[a]; ob[3 + a] = ob[3 + a] > q[a] ? ob[3 + a] : q[a]; }
        }
    }
}

int build_forest_host(const BuildInput& in, const InstancedScene& sc, const ForestLayout& L, int split_method, int max_prims_in_node, BuildOutput& out, std::vector<ForestTreeOut>& trees) {
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t n_trees = (uint32_t)L.tree_start.size() - 1u;
    trees.assign(n_trees, ForestTreeOut{});
    std::vector<BuildOutput> built(n_trees);
    // 1. the objects' aggregates
    for (uint32_t t = 1; t < n_trees; t++) {
        BuildInput oi = in; oi.items = L.items.data() + L.tree_start[t]; oi.n_items = L.tree_start[t + 1] - L.tree_start[t]; oi.inst_bounds = nullptr;
        const int brc = build_bvh(oi, split_method, max_prims_in_node, 0, built[t]);
        if (brc != 0) return brc;
    }
    // 2. TransformedPrimitive::world_bound of every instance, then the scene's aggregate
    std::vector<float> ibounds(6 * sc.n_inst);
    for (size_t k = 0; k < sc.n_inst; k++) transform_bounds(sc.inst_i2w + 16 * k, built[L.inst_tree[k]].root_lo, built[L.inst_tree[k]].root_hi, &ibounds[6 * k]);
    {
        BuildInput ti = in; ti.items = L.items.data(); ti.n_items = L.tree_start[1]; ti.inst_bounds = ibounds.data();
        const int brc = build_bvh(ti, split_method, max_prims_in_node, 0, built[0]);
        if (brc != 0) return brc;
    }
    // 3. one node array, one TriRec array
    out = BuildOutput();
    for (uint32_t t = 0; t < n_trees; t++) {
        const BuildOutput& bo = built[t];
        const uint32_t node_off = (uint32_t)out.nodes.size(), tri_off = (uint32_t)out.tris.size();
        auto fix = [&](uint32_t ref) { return (ref & PH_LEAF_BIT) ? (PH_LEAF_BIT | ((ref & ~PH_LEAF_BIT) + tri_off)) : ref + node_off; };
        for (Node64 nd : bo.nodes) { nd.c0 = fix(nd.c0); nd.c1 = fix(nd.c1); out.nodes.push_back(nd); }
        out.tris.insert(out.tris.end(), bo.tris.begin(), bo.tris.end());
        ForestTreeOut& fo = trees[t];
        fo.root_ref = fix(bo.root_ref); fo.n_items = L.tree_start[t + 1] - L.tree_start[t];
        for (int k = 0; k < 3; k++) { fo.lo[k] = bo.root_lo[k]; fo.hi[k] = bo.root_hi[k]; }
        out.interior_nodes += bo.interior_nodes; out.leaf_nodes += bo.leaf_nodes;
        out.max_leaf_prims = std::max(out.max_leaf_prims, bo.max_leaf_prims); out.max_depth = std::max(out.max_depth, bo.max_depth);
    }
    out.root_ref = trees[0].root_ref;
    for (int k = 0; k < 3; k++) { out.root_lo[k] = trees[0].lo[k]; out.root_hi[k] = trees[0].hi[k]; }
    out.total_nodes = out.interior_nodes + out.leaf_nodes;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

}  // namespace phost
