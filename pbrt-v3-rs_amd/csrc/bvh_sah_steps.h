// The per-item steps of the level-synchronous SAH build (bvh_sah_device.hip): BVHAccel::new with SplitMethod::SAH, the reference's default
// (accelerators/src/bvh/mod.rs:43-153, sah.rs:26-367), as data-parallel passes over one level of the tree at a time.  The tree is the one the
// reference's recursion makes — same buckets, same costs, same primitive order after every partition — so it does not matter where it was built.
//
// What makes that possible:
//  * a node's decision depends on its own primitive range only, so all nodes of a level can be decided at once;
//  * boxes are unions (min / max are exact: any grouping gives the same numbers), so a node's bound and centroid bound are the unions of the parent's
//    bucket boxes on its side of the split — the bucket pass also accumulates the centroid bounds per bucket, and no pass re-reads a child's primitives
//    for its bounds;
//  * `itertools::partition` (the two-pointer swap sah.rs:352-361 reorders the primitives with) has a closed form: with P passing elements, the k-th
//    failing element of the first P positions (ascending) changes places with the k-th passing element behind them (descending), everything else stays.
//    The ranks come from one prefix sum of the predicate over the whole array.
// One quantity is not reproduced: the SIGN of a zero box coordinate when a box's planes meet +0.0 and -0.0 (the reference's `<`-based min keeps whichever
// came first, an atomic min on the ordered bit pattern keeps -0.0).  Nothing downstream can tell: the planes are only subtracted from and compared.
//
// Every function here is one work item of one pass; the kernels of bvh_sah_device.hip run them over a grid.
#pragma once
#include "scene_types.h"
#include <cstdint>
#include <cstring>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace phs {

#define PHS_NONE 0xFFFFFFFFu
#define PHS_BUCKETS 12
#define PHS_BW 13                       // words per bucket: count, box lo xyz, box hi xyz, centroid lo xyz, centroid hi xyz (floats as ordered uints)
#define PHS_SLOT (PHS_BUCKETS * PHS_BW) // words of one node's bucket scratch
#define PHS_CHUNK 2048u                 // elements one block owns in the element passes

#if defined(__HIPCC__)
#define PHS_HD __host__ __device__ inline
#else
#define PHS_HD inline
#endif

PHS_HD float fmn(float a, float b) { return a < b ? a : b; }   // core/src/pbrt/common.rs:81-92 (`<`-based)
PHS_HD float fmx(float a, float b) { return a > b ? a : b; }
PHS_HD uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PHS_HD float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
// order-preserving map float -> uint32 for atomic min / max
PHS_HD uint32_t f2ord(float f) { const uint32_t u = fbits(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
PHS_HD float ord2f(uint32_t o) { return bitsf((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }
#define PHS_FLT_MAX 3.402823466e+38f

// the three read-modify-write operations of the passes (the host versions exist for the single-threaded rehearsal of the passes in scripts/sah_steps_check.cpp)
PHS_HD uint32_t a_add(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    const uint32_t o = *p; *p = o + v; return o;
#endif
}
PHS_HD void a_min(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
PHS_HD void a_max(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}

struct alignas(16) Elem { float x, y, z; uint32_t w; };   // one half of a primitive's record: a box corner + (lo half) the primitive's id

struct alignas(16) SNode {   // a build node: 96 B
    float lo[3]; uint32_t start;     // bound (bvh/common.rs:101-115) and primitive range [start, end)
    float hi[3]; uint32_t end;
    float clo[3]; uint32_t kid0;     // centroid bound; children kid0, kid0 + 1 — PHS_NONE: a leaf (or not decided yet)
    float chi[3]; uint32_t axis;     // split axis
    uint32_t slot;                   // this level's bucket scratch, PHS_NONE when the node needs none (leaf, or split without buckets)
    uint32_t best;                   // split_sah's min_cost_split_bucket
    uint32_t mid;                    // first position of the second child
    uint32_t size;                   // interior nodes in the subtree
    uint32_t A, index;               // numbering (see number_node)
    uint32_t pad[2];
};

// A FOREST is built in one go: tree t owns the positions [tree_start[t], tree_start[t + 1]) and starts as build node t.  Scenes with object instances use it — tree 0 is the scene's
// aggregate over its own triangles and its TransformedPrimitives, trees 1.. are the aggregates of the instanced objects (one accelerator per object, api/src/lib.rs:953-971) — and every
// tree's nodes are numbered after the previous tree's, so the arrays come out as the host builder lays them out: [scene | object | object ..].  A single tree is a forest of one.
// counters[]: 0 nodes made, 1 bucket slots handed out this level, 2 leaves, 3 largest leaf (4..15 spare)
struct Ctx {
    uint32_t n, max_prims;
    const float* P; const uint32_t* idx; const uint32_t* tri_flags; const uint32_t* tri_mesh;
    const uint32_t* items;           // per position before the build: triangle id, or PH_ITEM_INST | k (bounds in inst_bounds[6k..]); null = the triangles 0 .. n-1
    const float* inst_bounds;
    const uint32_t* tree_start;      // n_trees + 1 entries
    uint32_t n_trees;
    uint32_t* root_words;            // 12 per tree: bound and centroid bound of the tree's primitives as ordered uints
    Elem* e_lo; Elem* e_hi;          // the primitives in their current order: box corners, id in e_lo[i].w  (BVHPrimitiveInfo, bvh/common.rs:62-91)
    uint32_t* seg;                   // per POSITION: the build node whose range holds it, PHS_NONE once that node is a leaf
    uint8_t* bkt;                    // per position: the bucket of the primitive there (this level)
    SNode* nodes;
    uint32_t* counters;
    uint32_t* scratch;               // [slots][PHS_BUCKETS][PHS_BW]
    uint32_t* scan;                  // exclusive prefix sum of the partition predicate
    uint32_t* lf; uint32_t* lb;      // per segment, from its start: positions of the k-th misplaced failing / passing element
    uint32_t* leaf_last;
    Node64* out_nodes; TriRec* out_tris;
};

PHS_HD float box_area(const float* lo, const float* hi) {   // bounds3.rs:95-107
    if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return 0.0f;
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float h = dx * dy + dx * dz + dy * dz;
    return h + h;
}
PHS_HD int box_widest(const float* lo, const float* hi) {   // bounds3.rs:122-134
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
}
PHS_HD uint32_t bucket_of(float clo, float chi, float c) {   // sah.rs:309-313 (saturating `as usize`)
    float o = c - clo;
    if (chi > clo) o = o / (chi - clo);
    const float f = (float)PHS_BUCKETS * o;
    uint32_t b = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f);
    if (b == (uint32_t)PHS_BUCKETS) b = PHS_BUCKETS - 1;
    return b < (uint32_t)PHS_BUCKETS ? b : (uint32_t)(PHS_BUCKETS - 1);   // (an index past the buckets panics in the reference; it cannot arise from finite centroids)
}
PHS_HD float axis_of(const Elem& e, int k) { return k == 0 ? e.x : (k == 1 ? e.y : e.z); }
PHS_HD float centroid(const Elem& lo, const Elem& hi, int k) { return 0.5f * (axis_of(lo, k) + axis_of(hi, k)); }   // bvh/common.rs:74-80

#define PHS_ITEM_INST 0x80000000u      // = PH_ITEM_INST (bvh_build.h)
PHS_HD uint32_t tree_of(const Ctx& c, uint32_t i) {   // the tree whose range holds position i
    uint32_t lo = 0u, hi = c.n_trees;
    while (hi - lo > 1u) { const uint32_t m = (lo + hi) >> 1; if (c.tree_start[m] <= i) lo = m; else hi = m; }
    return lo;
}
// ---- pass 0: Triangle::world_bound (triangle.rs:427-431) — or the TransformedPrimitive's, handed in — per primitive, and its tree's two bounds into dst[12] --------------
PHS_HD void init_elem(const Ctx& c, uint32_t i, uint32_t tree, uint32_t* dst) {
    const uint32_t id = c.items ? c.items[i] : i;
    float lo[3], hi[3];
    if (id & PHS_ITEM_INST) {
        const float* bb = c.inst_bounds + 6 * (size_t)(id & ~PHS_ITEM_INST);
        for (int k = 0; k < 3; k++) { lo[k] = bb[k]; hi[k] = bb[3 + k]; }
    } else {
        const float* a = c.P + 3 * (size_t)c.idx[3 * (size_t)id];
        const float* b = c.P + 3 * (size_t)c.idx[3 * (size_t)id + 1];
        const float* d = c.P + 3 * (size_t)c.idx[3 * (size_t)id + 2];
        for (int k = 0; k < 3; k++) { lo[k] = fmn(fmn(a[k], b[k]), d[k]); hi[k] = fmx(fmx(a[k], b[k]), d[k]); }
    }
    Elem l, h;
    l.x = lo[0]; l.y = lo[1]; l.z = lo[2]; l.w = id;
    h.x = hi[0]; h.y = hi[1]; h.z = hi[2]; h.w = 0u;
    c.e_lo[i] = l; c.e_hi[i] = h; c.seg[i] = tree;
    for (int k = 0; k < 3; k++) {
        const float ctr = 0.5f * (lo[k] + hi[k]);
        a_min(dst + k, f2ord(lo[k])); a_max(dst + 3 + k, f2ord(hi[k]));
        a_min(dst + 6 + k, f2ord(ctr)); a_max(dst + 9 + k, f2ord(ctr));
    }
}
PHS_HD void init_bounds_words(uint32_t* dst) {   // an empty Bounds3f (bounds3.rs:23-30) as ordered uints, twice
    for (int k = 0; k < 3; k++) { dst[k] = f2ord(PHS_FLT_MAX); dst[3 + k] = f2ord(-PHS_FLT_MAX); dst[6 + k] = f2ord(PHS_FLT_MAX); dst[9 + k] = f2ord(-PHS_FLT_MAX); }
}
PHS_HD void make_root(const Ctx& c, uint32_t t) {   // build node t = the root of tree t; its numbers (A, index) are set once the trees' sizes are known (root_numbers)
    SNode r;
    const uint32_t* g = c.root_words + 12u * t;
    for (int k = 0; k < 3; k++) { r.lo[k] = ord2f(g[k]); r.hi[k] = ord2f(g[3 + k]); r.clo[k] = ord2f(g[6 + k]); r.chi[k] = ord2f(g[9 + k]); }
    r.start = c.tree_start[t]; r.end = c.tree_start[t + 1u]; r.kid0 = PHS_NONE; r.axis = 0u; r.slot = PHS_NONE; r.best = 0u; r.mid = 0u; r.size = 0u; r.A = 1u; r.index = 0u; r.pad[0] = r.pad[1] = 0u;
    c.nodes[t] = r;
    if (t == 0u) c.counters[0] = c.n_trees;
}
// tree t's interior nodes take the slots [base, base + size): base = the interior nodes of the trees before it
PHS_HD void root_numbers(const Ctx& c, uint32_t t, uint32_t base) { c.nodes[t].index = base; c.nodes[t].A = base + 1u; }

PHS_HD void count_leaf(const Ctx& c, uint32_t n_prims) { a_add(c.counters + 2, 1u); a_max(c.counters + 3, n_prims); }

// ---- pass 1, per node of the level: the decisions that need no bucket (sah.rs:44-63, :240-254) --------------------------------------------------------
PHS_HD void decide_node(const Ctx& c, uint32_t v) {
    SNode& nd = c.nodes[v];
    const uint32_t n = nd.end - nd.start;
    nd.kid0 = PHS_NONE; nd.slot = PHS_NONE;
    if (n == 1u) { count_leaf(c, 1u); return; }
    const int dim = box_widest(nd.clo, nd.chi);
    if (nd.chi[dim] == nd.clo[dim]) { count_leaf(c, n); return; }   // all centroids in one place (sah.rs:61-63)
    nd.axis = (uint32_t)dim;
    if (n == 2u) {
        // split_equal_counts on two primitives: every correct selection leaves [smaller, larger]
        Elem l0 = c.e_lo[nd.start], h0 = c.e_hi[nd.start], l1 = c.e_lo[nd.start + 1u], h1 = c.e_hi[nd.start + 1u];
        if (centroid(l1, h1, dim) < centroid(l0, h0, dim)) {
            c.e_lo[nd.start] = l1; c.e_hi[nd.start] = h1; c.e_lo[nd.start + 1u] = l0; c.e_hi[nd.start + 1u] = h0;
            const Elem tl = l0, th = h0; l0 = l1; h0 = h1; l1 = tl; h1 = th;
        }
        const uint32_t k = a_add(c.counters, 2u);
        nd.kid0 = k; nd.mid = nd.start + 1u;
        for (int s = 0; s < 2; s++) {
            const Elem& l = s ? l1 : l0; const Elem& h = s ? h1 : h0;
            SNode kd;
            kd.lo[0] = l.x; kd.lo[1] = l.y; kd.lo[2] = l.z; kd.hi[0] = h.x; kd.hi[1] = h.y; kd.hi[2] = h.z;
            for (int q = 0; q < 3; q++) kd.clo[q] = kd.chi[q] = centroid(l, h, q);
            kd.start = nd.start + (uint32_t)s; kd.end = kd.start + 1u; kd.kid0 = PHS_NONE; kd.axis = 0u; kd.slot = PHS_NONE; kd.best = 0u; kd.mid = 0u; kd.size = 0u; kd.A = 0u; kd.index = 0u;
            kd.pad[0] = kd.pad[1] = 0u;
            c.nodes[k + (uint32_t)s] = kd;
        }
        return;
    }
    nd.slot = a_add(c.counters + 1, 1u);
}

// ---- pass 2, per position: the primitive's bucket, and its boxes into the node's scratch (sah.rs:305-323) ---------------------------------------------
// dst: where this node's buckets are accumulated — the global scratch, or a block-local copy when the whole block works for one node
PHS_HD void bucket_elem(const Ctx& c, uint32_t i, uint32_t* block_dst) {
    const uint32_t v = c.seg[i];
    if (v == PHS_NONE) return;
    const SNode& nd = c.nodes[v];
    if (nd.slot == PHS_NONE) return;
    const Elem l = c.e_lo[i], h = c.e_hi[i];
    const int dim = (int)nd.axis;
    const uint32_t b = bucket_of(nd.clo[dim], nd.chi[dim], centroid(l, h, dim));
    c.bkt[i] = (uint8_t)b;
    uint32_t* d = (block_dst ? block_dst : c.scratch + (size_t)nd.slot * PHS_SLOT) + b * PHS_BW;
    a_add(d, 1u);
    a_min(d + 1, f2ord(l.x)); a_min(d + 2, f2ord(l.y)); a_min(d + 3, f2ord(l.z));
    a_max(d + 4, f2ord(h.x)); a_max(d + 5, f2ord(h.y)); a_max(d + 6, f2ord(h.z));
    for (int k = 0; k < 3; k++) { const uint32_t o = f2ord(centroid(l, h, k)); a_min(d + 7 + k, o); a_max(d + 10 + k, o); }
}
PHS_HD uint32_t scratch_init_word(uint32_t w) {   // word w of an empty slot
    const uint32_t f = w % PHS_BW;
    return f == 0u ? 0u : ((f <= 3u || (f >= 7u && f <= 9u)) ? f2ord(PHS_FLT_MAX) : f2ord(-PHS_FLT_MAX));
}
PHS_HD void scratch_merge_word(uint32_t* dst, uint32_t w, uint32_t val) {   // a block-local word into the global slot
    const uint32_t f = w % PHS_BW;
    if (f == 0u) { if (val) a_add(dst + w, val); }
    else if (f <= 3u || (f >= 7u && f <= 9u)) a_min(dst + w, val);
    else a_max(dst + w, val);
}

// ---- pass 3, per node with buckets: split_sah's cost loop and the leaf / split decision (sah.rs:325-367) --------------------------------------------------
PHS_HD void eval_node(const Ctx& c, uint32_t v) {
    SNode& nd = c.nodes[v];
    if (nd.slot == PHS_NONE) return;
    const uint32_t* s = c.scratch + (size_t)nd.slot * PHS_SLOT;
    const uint32_t n = nd.end - nd.start;
    uint32_t cnt[PHS_BUCKETS];
    float blo[PHS_BUCKETS][3], bhi[PHS_BUCKETS][3];
    for (int b = 0; b < PHS_BUCKETS; b++) {
        cnt[b] = s[b * PHS_BW];
        for (int k = 0; k < 3; k++) { blo[b][k] = ord2f(s[b * PHS_BW + 1 + k]); bhi[b][k] = ord2f(s[b * PHS_BW + 4 + k]); }
    }
    // suffix unions once, the prefix grown on the way (min / max are exact, the grouping does not matter)
    float slo[PHS_BUCKETS][3], shi[PHS_BUCKETS][3]; uint32_t sc[PHS_BUCKETS];
    for (int k = 0; k < 3; k++) { slo[PHS_BUCKETS - 1][k] = blo[PHS_BUCKETS - 1][k]; shi[PHS_BUCKETS - 1][k] = bhi[PHS_BUCKETS - 1][k]; }
    sc[PHS_BUCKETS - 1] = cnt[PHS_BUCKETS - 1];
    for (int b = PHS_BUCKETS - 2; b >= 0; b--) {
        for (int k = 0; k < 3; k++) { slo[b][k] = fmn(slo[b + 1][k], blo[b][k]); shi[b][k] = fmx(shi[b + 1][k], bhi[b][k]); }
        sc[b] = sc[b + 1] + cnt[b];
    }
    float plo[3] = {PHS_FLT_MAX, PHS_FLT_MAX, PHS_FLT_MAX}, phi[3] = {-PHS_FLT_MAX, -PHS_FLT_MAX, -PHS_FLT_MAX};
    uint32_t pc = 0u;
    const float whole = box_area(nd.lo, nd.hi);
    float best = 0.0f; int best_b = 0;
    for (int b = 0; b < PHS_BUCKETS - 1; b++) {
        for (int k = 0; k < 3; k++) { plo[k] = fmn(plo[k], blo[b][k]); phi[k] = fmx(phi[k], bhi[b][k]); }
        pc += cnt[b];
        const float cost = 1.0f + ((float)pc * box_area(plo, phi) + (float)sc[b + 1] * box_area(slo[b + 1], shi[b + 1])) / whole;
        if (b == 0 || cost < best) { best = cost; best_b = b; }
    }
    const float leaf_cost = (float)n;
    uint32_t passing = 0u;
    for (int b = 0; b <= best_b; b++) passing += cnt[b];
    if (!(n > c.max_prims || best < leaf_cost) || passing == 0u || passing == n) {   // a leaf (the reference panics on an empty side, sah.rs:37: a leaf here)
        nd.slot = PHS_NONE;
        count_leaf(c, n);
        return;
    }
    const uint32_t k = a_add(c.counters, 2u);
    nd.kid0 = k; nd.best = (uint32_t)best_b; nd.mid = nd.start + passing;
    for (int side = 0; side < 2; side++) {
        SNode kd;
        for (int q = 0; q < 3; q++) { kd.lo[q] = PHS_FLT_MAX; kd.hi[q] = -PHS_FLT_MAX; kd.clo[q] = PHS_FLT_MAX; kd.chi[q] = -PHS_FLT_MAX; }
        const int b0 = side ? best_b + 1 : 0, b1 = side ? PHS_BUCKETS - 1 : best_b;
        for (int b = b0; b <= b1; b++)
            for (int q = 0; q < 3; q++) {
                kd.lo[q] = fmn(kd.lo[q], blo[b][q]); kd.hi[q] = fmx(kd.hi[q], bhi[b][q]);
                kd.clo[q] = fmn(kd.clo[q], ord2f(s[b * PHS_BW + 7 + q])); kd.chi[q] = fmx(kd.chi[q], ord2f(s[b * PHS_BW + 10 + q]));
            }
        kd.start = side ? nd.mid : nd.start; kd.end = side ? nd.end : nd.mid;
        kd.kid0 = PHS_NONE; kd.axis = 0u; kd.slot = PHS_NONE; kd.best = 0u; kd.mid = 0u; kd.size = 0u; kd.A = 0u; kd.index = 0u; kd.pad[0] = kd.pad[1] = 0u;
        c.nodes[k + (uint32_t)side] = kd;
    }
}

// ---- pass 4: the partition (sah.rs:352-361 = itertools::partition) in closed form ------------------------------------------------------------------------------
// the predicate `bucket <= min_cost_split_bucket` at position i, 0 for positions whose node does not partition this level
PHS_HD uint32_t flag_of(const Ctx& c, uint32_t i) {
    const uint32_t v = c.seg[i];
    if (v == PHS_NONE) return 0u;
    const SNode& nd = c.nodes[v];
    if (nd.slot == PHS_NONE) return 0u;
    return (uint32_t)c.bkt[i] <= nd.best ? 1u : 0u;
}
PHS_HD void list_elem(const Ctx& c, uint32_t i) {
    const uint32_t v = c.seg[i];
    if (v == PHS_NONE) return;
    const SNode& nd = c.nodes[v];
    if (nd.slot == PHS_NONE) return;
    const uint32_t passing = nd.mid - nd.start, before = c.scan[i] - c.scan[nd.start], li = i - nd.start;
    const bool pass = (uint32_t)c.bkt[i] <= nd.best;
    if (li < passing && !pass) c.lf[nd.start + (li - before)] = i;                     // the (li - before)-th failing element of the front part
    else if (li >= passing && pass) c.lb[nd.start + (passing - before - 1u)] = i;      // as many passing elements behind it as its rank from the back
}
PHS_HD void swap_pos(const Ctx& c, uint32_t j) {
    const uint32_t v = c.seg[j];
    if (v == PHS_NONE) return;
    const SNode& nd = c.nodes[v];
    if (nd.slot == PHS_NONE) return;
    const uint32_t passing = nd.mid - nd.start;
    const uint32_t misplaced = passing - (c.scan[nd.mid] - c.scan[nd.start]);   // failing elements among the first `passing` positions
    if (j - nd.start >= misplaced) return;
    const uint32_t a = c.lf[j], b = c.lb[j];
    const Elem la = c.e_lo[a], ha = c.e_hi[a], lb_ = c.e_lo[b], hb = c.e_hi[b];
    c.e_lo[a] = lb_; c.e_hi[a] = hb; c.e_lo[b] = la; c.e_hi[b] = ha;
}
PHS_HD void relabel_pos(const Ctx& c, uint32_t i) {
    const uint32_t v = c.seg[i];
    if (v == PHS_NONE) return;
    const SNode& nd = c.nodes[v];
    c.seg[i] = nd.kid0 == PHS_NONE ? PHS_NONE : (i < nd.mid ? nd.kid0 : nd.kid0 + 1u);
}

// ---- after the last level: sizes bottom-up, numbers top-down, records --------------------------------------------------------------------------------------------
PHS_HD void size_node(const Ctx& c, uint32_t v) {
    SNode& nd = c.nodes[v];
    nd.size = nd.kid0 == PHS_NONE ? 0u : 1u + c.nodes[nd.kid0].size + c.nodes[nd.kid0 + 1u].size;
    if (nd.kid0 == PHS_NONE) c.leaf_last[nd.end - 1u] = 1u;
}
// The host builder's numbering (bvh_build.cpp: a pre-order walk that reserves both children's slots when it reaches their parent), without the walk:
// `A` = slots handed out when the walk reaches the node.  The children take A and A + 1 (interior ones only); the walk then descends into child 0 with
// A + (interior children), and reaches child 1 after child 0's whole subtree, which hands out one slot per interior node below child 0.
PHS_HD void number_node(const Ctx& c, uint32_t v) {
    const SNode& nd = c.nodes[v];
    if (nd.kid0 == PHS_NONE) return;
    SNode& k0 = c.nodes[nd.kid0]; SNode& k1 = c.nodes[nd.kid0 + 1u];
    const uint32_t i0 = k0.kid0 != PHS_NONE ? 1u : 0u, i1 = k1.kid0 != PHS_NONE ? 1u : 0u;
    uint32_t r0 = PH_LEAF_BIT | k0.start, r1 = PH_LEAF_BIT | k1.start;
    if (i0) { k0.index = nd.A; k0.A = nd.A + i0 + i1; r0 = k0.index; }
    if (i1) { k1.index = nd.A + i0; k1.A = nd.A + i0 + i1 + (i0 ? k0.size - 1u : 0u); r1 = k1.index; }
    Node64 o;
    o.x0[0] = k0.lo[0]; o.x0[1] = k0.hi[0]; o.y0[0] = k0.lo[1]; o.y0[1] = k0.hi[1]; o.z0[0] = k0.lo[2]; o.z0[1] = k0.hi[2];
    o.x1[0] = k1.lo[0]; o.x1[1] = k1.hi[0]; o.y1[0] = k1.lo[1]; o.y1[1] = k1.hi[1]; o.z1[0] = k1.lo[2]; o.z1[1] = k1.hi[2];
    o.c0 = r0; o.c1 = r1; o.axis = nd.axis; o.pad = 0u;
    c.out_nodes[nd.index] = o;
}
PHS_HD void emit_tri(const Ctx& c, uint32_t i) {
    const uint32_t id = c.e_lo[i].w;
    if (id & PHS_ITEM_INST) {   // a TransformedPrimitive: the 48-byte record names the instance (bvh_build.cpp does the same)
        TriRec r;
        memset(&r, 0, sizeof r);
        r.prim = id & ~PHS_ITEM_INST; r.flags = PH_TRI_INSTANCE | (c.leaf_last[i] ? PH_TRI_LAST : 0u);
        c.out_tris[i] = r;
        return;
    }
    const float* p0 = c.P + 3 * (size_t)c.idx[3 * (size_t)id]; const float* p1 = c.P + 3 * (size_t)c.idx[3 * (size_t)id + 1]; const float* p2 = c.P + 3 * (size_t)c.idx[3 * (size_t)id + 2];
    TriRec r;
    r.p0[0] = p0[0]; r.p0[1] = p0[1]; r.p0[2] = p0[2]; r.prim = id;
    r.p1[0] = p1[0]; r.p1[1] = p1[1]; r.p1[2] = p1[2]; r.flags = ((c.tri_flags ? c.tri_flags[id] : 0u) & ~PH_TRI_LAST) | (c.leaf_last[i] ? PH_TRI_LAST : 0u);
    r.p2[0] = p2[0]; r.p2[1] = p2[1]; r.p2[2] = p2[2]; r.mesh = c.tri_mesh ? c.tri_mesh[id] : 0u;
    c.out_tris[i] = r;
}

}  // namespace phs
