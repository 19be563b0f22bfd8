// HLBVH construction on the device: BVHAccel::new with SplitMethod::HLBVH (accelerators/src/bvh/hlbvh.rs:33-449, morton.rs:33-120),
// the same tree the host builder (bvh_build.cpp) and the reference make — including the reference's Morton quirk (quirk B10: the code interleaves the
// low bits of the IEEE BIT PATTERN of the scaled centroid offset, morton.rs:33-39) — so closest hits, ties included, do not depend on where the tree was built.
//
//   K1  primitive bounds + scene bounds        compute_morton_primitives' inputs (hlbvh.rs:52-60; Triangle::world_bound triangle.rs:427-431)
//   K2  Morton codes                           hlbvh.rs:97-135, morton.rs:33-48, :101-118
//   K3  stable LSD radix sort, 4 x 8 bits      morton.rs:50-98 sorts 5 x 6 bits; any stable sort by the 30-bit code gives the same order
//   K4  treelet ranges (top 12 code bits)      hlbvh.rs:62-84
//   K5  emit_lbvh, one level per launch        hlbvh.rs:199-294 (all treelets side by side; round 3 — rounds 1 - 2 gave each treelet one thread)
//   --  SAH over the <= 4096 treelet roots     hlbvh.rs:296-432, on the host (bvh_build.cpp: build_upper_sah)
//   K6  Node64 / TriRec emission               the device layout of scene_types.h; leaves in depth-first order as flatten_bvh_tree leaves them
//
// Inside a treelet emit_lbvh always splits a sorted range into a lower and an upper part, so the depth-first leaf order of a treelet IS the sorted
// order; the scene's leaf order is the treelets' ranges concatenated in the depth-first order of the upper SAH tree.
#include "scene_host.h"
#include "host_math.h"
#include <algorithm>
#include <chrono>
#include <cstring>

namespace phd {

struct DNode {  // a build node of a treelet: 48 B
    float lo[3]; uint32_t kid0;   // kids: pool indices; leaf: 0xFFFFFFFF
    float hi[3]; uint32_t kid1;
    uint32_t first, count;        // leaf: range of the SORTED primitive list
    uint32_t axis, dense;         // interior: split axis, index among the treelet's interior nodes in creation order
};
#define PHD_NONE 0xFFFFFFFFu

__device__ __forceinline__ float fmn(float a, float b) { return a < b ? a : b; }   // core/src/pbrt/common.rs:81-92 (`<`-based)
__device__ __forceinline__ float fmx(float a, float b) { return a > b ? a : b; }
// order-preserving map float -> uint32 for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t o) { const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f; }

// K1: one thread per triangle
__global__ __launch_bounds__(256) void prim_bounds_kernel(const float* P, const uint32_t* idx, uint32_t n, float* blo, float* bhi, uint32_t* gbounds /*6: ord(min xyz), ord(max xyz)*/) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    const bool ok = i < n;
    if (ok) {
        const float* a = P + 3 * (size_t)idx[3 * (size_t)i];
        const float* b = P + 3 * (size_t)idx[3 * (size_t)i + 1];
        const float* c = P + 3 * (size_t)idx[3 * (size_t)i + 2];
        for (int k = 0; k < 3; k++) {
            lo[k] = fmn(fmn(a[k], b[k]), c[k]); hi[k] = fmx(fmx(a[k], b[k]), c[k]);
            blo[3 * (size_t)i + k] = lo[k]; bhi[3 * (size_t)i + k] = hi[k];
        }
    }
    __shared__ uint32_t sm[6];
    if (threadIdx.x < 3) { sm[threadIdx.x] = 0xFFFFFFFFu; sm[3 + threadIdx.x] = 0u; }
    __syncthreads();
    if (ok) for (int k = 0; k < 3; k++) { atomicMin(&sm[k], f2ord(lo[k])); atomicMax(&sm[3 + k], f2ord(hi[k])); }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&gbounds[threadIdx.x], sm[threadIdx.x]); atomicMax(&gbounds[3 + threadIdx.x], sm[3 + threadIdx.x]); }
}

__device__ __forceinline__ uint32_t left_shift_3(uint32_t x) {  // morton.rs:101-118
    uint32_t v = (x == (1u << 10)) ? x - 1 : x;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// K2
__global__ __launch_bounds__(256) void morton_kernel(const float* blo, const float* bhi, uint32_t n, const uint32_t* gbounds, uint32_t* codes, uint32_t* ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3];
    for (int k = 0; k < 3; k++) {
        const float glo = ord2f(gbounds[k]), ghi = ord2f(gbounds[3 + k]);
        const float ctr = 0.5f * (blo[3 * (size_t)i + k] + bhi[3 * (size_t)i + k]);   // BVHPrimitiveInfo::new (bvh/common.rs:74-80)
        float o = ctr - glo;                                                          // Bounds3::offset (bounds3.rs:222-238)
        if (ghi > glo) o = o / (ghi - glo);
        c[k] = left_shift_3(__float_as_uint(o * 1024.0f));                            // morton.rs:33-39: the BITS of the float
    }
    codes[i] = (c[2] << 2) | (c[1] << 1) | c[0];
    ids[i] = i;
}

// K3: one pass of a stable least-significant-digit radix sort, 8-bit digits.  Block b owns the contiguous slice [b * per, (b + 1) * per).
#define PHD_RS_BLOCK 256
__global__ __launch_bounds__(PHD_RS_BLOCK) void rs_hist_kernel(const uint32_t* keys, uint32_t n, uint32_t per, int shift, uint32_t* block_hist /*[256][gridDim.x]*/) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PHD_RS_BLOCK) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    block_hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of `m` counters in place, one block
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t* a, uint32_t m) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (m + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0u;
    for (uint32_t i = lo; i < hi; i++) sum += a[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = a[i]; a[i] = run; run += v; }
}
__global__ __launch_bounds__(PHD_RS_BLOCK) void rs_scatter_kernel(const uint32_t* keys, const uint32_t* vals, uint32_t* keys_out, uint32_t* vals_out, uint32_t n, uint32_t per, int shift,
                                                                 const uint32_t* block_base /*scanned [256][gridDim.x]*/) {
    __shared__ uint32_t run[256];                       // where the block's next key of each digit goes
    __shared__ uint32_t cnt[PHD_RS_BLOCK / 64][256];    // this round's keys per wave and digit
    run[threadIdx.x] = block_base[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lane_lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t base = lo; base < hi; base += PHD_RS_BLOCK) {   // rounds of 256 keys, in order: stability
        for (uint32_t w = 0; w < PHD_RS_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t i = base + threadIdx.x;
        const bool ok = i < hi;
        uint32_t key = 0, val = 0, d = 0;
        if (ok) { key = keys[i]; val = vals[i]; d = (key >> shift) & 255u; }
        // lanes of the wave with the same digit (eight ballots), rank among them
        uint64_t same = __ballot(ok);
        for (int b = 0; b < 8; b++) { const uint64_t m = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? m : ~m; }
        const uint32_t rank = (uint32_t)__popcll(same & lane_lt);
        if (ok && rank == 0u) cnt[wave][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (ok) {
            uint32_t pos = run[d] + rank;
            for (uint32_t w = 0; w < wave; w++) pos += cnt[w][d];
            keys_out[pos] = key; vals_out[pos] = val;
        }
        __syncthreads();
        uint32_t tot = 0u;
        for (uint32_t w = 0; w < PHD_RS_BLOCK / 64; w++) tot += cnt[w][threadIdx.x];
        run[threadIdx.x] += tot;
        __syncthreads();
    }
}

// K4: first sorted index whose treelet key (code bits 18..29) is >= k, for k = 0 .. 4096
__global__ void treelet_start_kernel(const uint32_t* codes, uint32_t n, uint32_t* start /*4097*/) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 4096u) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((codes[mid] >> 18) & 0xFFFu) < k) lo = mid + 1; else hi = mid; }
    start[k] = lo;
}

// K5: emit_lbvh (hlbvh.rs:199-294) for all treelets at once, one LEVEL of the recursion per launch (round 3).  The reference's Morton quirk (B10) leaves a few treelets with
// millions of primitives; one thread per treelet — rounds 1 - 2 — spent 0.3 s of a 10 M-triangle build walking those alone.  emit_lbvh only ever cuts a sorted range in two
// at the first index whose code differs in the current bit, so a node is known by its range: every node of a level is decided independently (skip the bits that do not
// split, leaf or binary search), and what the recursion's order decides — the interior nodes' creation (pre-order) numbers, the boxes — follows from two more sweeps over the
// levels: bottom-up the boxes and the number of interior nodes below each node, top-down `dense` = the parent's number + 1 (+ the first child's subtree for the second child).
// A node's pool slot is a function of its range (interior: 2 x its split position; leaf: 2 x its first position + 1; a treelet's root: 2 x the treelet's first position,
// where the host looks for it), so no slot counter is shared.
struct TreeletInfo { uint32_t first, n, interior, leaves, max_leaf, depth, pad[2]; };
struct EmitItem { uint32_t first, n, slot_of_parent, tree; int bit; uint32_t which, self, pad; };   // which: 0 / 1 = first / second child, 2 = a treelet's root
#define PHD_MAX_LEVELS 20   // a split consumes at least one of the 18 code bits below the treelet key: 19 levels at most

__global__ __launch_bounds__(256) void emit_roots_kernel(const uint32_t* tl_first, const uint32_t* tl_n, uint32_t n_treelets, EmitItem* items, uint32_t* lvl /*[PHD_MAX_LEVELS + 2], [0] = 0*/, uint32_t* n_items,
                                                         TreeletInfo* info) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { lvl[0] = 0u; lvl[1] = n_treelets; *n_items = n_treelets; }
    if (t >= n_treelets) return;
    items[t] = EmitItem{tl_first[t], tl_n[t], PHD_NONE, t, 29 - 12, 2u, 0u, 0u};
    info[t] = TreeletInfo{tl_first[t], tl_n[t], 0u, 0u, 0u, 0u, {0u, 0u}};
}
// one level: every item decides leaf / split, writes its node, tells its parent where it lives and appends its two children to the next level
__global__ __launch_bounds__(256) void emit_level_kernel(EmitItem* items, const uint32_t* lvl, int level, uint32_t* n_items, const uint32_t* codes, uint32_t max_prims, DNode* pool, uint32_t* leaf_last,
                                                         TreeletInfo* info) {
    const uint32_t i = lvl[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lvl[level + 1]) return;
    EmitItem it = items[i];
    int bit = it.bit;
    bool leaf = false;
    for (;;) {
        if (bit == -1 || it.n < max_prims) { leaf = true; break; }
        const uint32_t mask = 1u << bit;
        if ((codes[it.first] & mask) != (codes[it.first + it.n - 1] & mask)) break;
        bit--;   // no split on this bit
    }
    uint32_t hi = 0;
    if (!leaf) {   // hlbvh.rs:253-268: first index whose bit differs from the first primitive's
        const uint32_t mask = 1u << bit;
        uint32_t lo = 0; hi = it.n - 1;
        while (lo + 1 != hi) {
            const uint32_t mid = (lo + hi) / 2;
            if ((codes[it.first + lo] & mask) == (codes[it.first + mid] & mask)) lo = mid; else hi = mid;
        }
    }
    const uint32_t self = it.which == 2u ? 2u * it.first : (leaf ? 2u * it.first + 1u : 2u * (it.first + hi));
    items[i].self = self; items[i].bit = leaf ? -2 : bit;   // (-2: a leaf, for the two sweeps that follow)
    if (it.which == 0u) pool[it.slot_of_parent].kid0 = self; else if (it.which == 1u) pool[it.slot_of_parent].kid1 = self;
    DNode& nd = pool[self];
    if (leaf) {
        nd.kid0 = nd.kid1 = PHD_NONE; nd.first = it.first; nd.count = it.n; nd.axis = 0; nd.dense = 0;
        leaf_last[it.first + it.n - 1] = 1u;
        atomicAdd(&info[it.tree].leaves, 1u); atomicMax(&info[it.tree].max_leaf, it.n); atomicMax(&info[it.tree].depth, (uint32_t)level + 1u);
        return;
    }
    nd.count = 0; nd.first = 0; nd.axis = (uint32_t)(bit % 3); nd.dense = 0;
    const uint32_t at = atomicAdd(n_items, 2u);
    items[at] = EmitItem{it.first, hi, self, it.tree, bit - 1, 0u, 0u, 0u};
    items[at + 1u] = EmitItem{it.first + hi, it.n - hi, self, it.tree, bit - 1, 1u, 0u, 0u};
}
__global__ void emit_close_level_kernel(uint32_t* lvl, int level, const uint32_t* n_items) { if (threadIdx.x == 0 && blockIdx.x == 0) lvl[level + 2] = *n_items; }
// bottom-up: boxes, and in `first` of an interior node the number of interior nodes of its subtree
__global__ __launch_bounds__(256) void emit_up_kernel(const EmitItem* items, const uint32_t* lvl, int level, const uint32_t* ids, const float* blo, const float* bhi, DNode* pool) {
    const uint32_t i = lvl[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lvl[level + 1]) return;
    const EmitItem it = items[i];
    DNode& nd = pool[it.self];
    if (it.bit == -2) {
        float lo[3], hi[3];
        for (uint32_t k = 0; k < it.n; k++) {
            const size_t id = ids[it.first + k];
            for (int q = 0; q < 3; q++) {
                const float l = blo[3 * id + q], h = bhi[3 * id + q];
                lo[q] = k == 0 ? l : fmn(lo[q], l); hi[q] = k == 0 ? h : fmx(hi[q], h);
            }
        }
        for (int q = 0; q < 3; q++) { nd.lo[q] = lo[q]; nd.hi[q] = hi[q]; }
        return;
    }
    const DNode& a = pool[nd.kid0]; const DNode& b = pool[nd.kid1];
    for (int q = 0; q < 3; q++) { nd.lo[q] = fmn(a.lo[q], b.lo[q]); nd.hi[q] = fmx(a.hi[q], b.hi[q]); }
    nd.first = 1u + (a.kid0 == PHD_NONE ? 0u : a.first) + (b.kid0 == PHD_NONE ? 0u : b.first);
}
// top-down: the creation numbers of the recursion (a node, then its whole first subtree, then the second)
__global__ __launch_bounds__(256) void emit_down_kernel(const EmitItem* items, const uint32_t* lvl, int level, DNode* pool, TreeletInfo* info) {
    const uint32_t i = lvl[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lvl[level + 1]) return;
    const EmitItem it = items[i];
    if (it.bit == -2) return;
    DNode& nd = pool[it.self];
    if (it.which == 2u) { nd.dense = 0u; info[it.tree].interior = nd.first; }
    DNode& a = pool[nd.kid0]; DNode& b = pool[nd.kid1];
    const uint32_t below_a = a.kid0 == PHD_NONE ? 0u : a.first;
    if (a.kid0 != PHD_NONE) a.dense = nd.dense + 1u;
    if (b.kid0 != PHD_NONE) b.dense = nd.dense + 1u + below_a;
}

// K6a: every interior build node of every treelet -> its Node64 (both children's boxes, child references in the final numbering)
__global__ __launch_bounds__(256) void convert_kernel(const DNode* pool, const uint32_t* tl_first, const uint32_t* tl_n, const uint32_t* tl_dense_base, const uint32_t* tl_out_base, uint32_t n_treelets,
                                                      Node64* nodes) {
    const uint32_t t = blockIdx.y;
    if (t >= n_treelets) return;
    const uint32_t T0 = tl_first[t], used = 2u * tl_n[t] - 1u;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < used; j += gridDim.x * blockDim.x) {
        const DNode nd = pool[2u * T0 + j];
        if (nd.kid0 == PHD_NONE) continue;   // a leaf, or a slot the treelet did not use (pool is cleared to 0xFF first)
        const DNode a = pool[nd.kid0], b = pool[nd.kid1];
        auto ref = [&](const DNode& c) { return c.kid0 == PHD_NONE ? (PH_LEAF_BIT | (tl_out_base[t] + (c.first - T0))) : tl_dense_base[t] + c.dense; };
        Node64 o;
        o.x0[0] = a.lo[0]; o.x0[1] = a.hi[0]; o.y0[0] = a.lo[1]; o.y0[1] = a.hi[1]; o.z0[0] = a.lo[2]; o.z0[1] = a.hi[2];
        o.x1[0] = b.lo[0]; o.x1[1] = b.hi[0]; o.y1[0] = b.lo[1]; o.y1[1] = b.hi[1]; o.z1[0] = b.lo[2]; o.z1[1] = b.hi[2];
        o.c0 = ref(a); o.c1 = ref(b); o.axis = nd.axis; o.pad = 0;
        nodes[tl_dense_base[t] + nd.dense] = o;
    }
}
// K6b: leaf records in the final order
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* ids, const uint32_t* leaf_last, const uint32_t* codes, const uint32_t* start4097, const uint32_t* key_out_base /*4096: final offset of the
                                                     treelet with that key*/, uint32_t n, const float* P, const uint32_t* idx, const uint32_t* tri_flags, const uint32_t* tri_mesh, TriRec* tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = (codes[i] >> 18) & 0xFFFu;
    const uint32_t q = key_out_base[key] + (i - start4097[key]);
    const uint32_t id = ids[i];
    const float* p0 = P + 3 * (size_t)idx[3 * (size_t)id]; const float* p1 = P + 3 * (size_t)idx[3 * (size_t)id + 1]; const float* p2 = P + 3 * (size_t)idx[3 * (size_t)id + 2];
    TriRec r;
    r.p0[0] = p0[0]; r.p0[1] = p0[1]; r.p0[2] = p0[2]; r.prim = id;
    r.p1[0] = p1[0]; r.p1[1] = p1[1]; r.p1[2] = p1[2]; r.flags = ((tri_flags ? tri_flags[id] : 0u) & ~PH_TRI_LAST) | (leaf_last[i] ? PH_TRI_LAST : 0u);
    r.p2[0] = p2[0]; r.p2[1] = p2[1]; r.p2[2] = p2[2]; r.mesh = tri_mesh ? tri_mesh[id] : 0u;
    tris[q] = r;
}

}  // namespace phd

namespace phost {

#define PHD_CHECK(call)                                                                 \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e__); (void)hipGetLastError(); goto fail; } } while (0)

// Builds the HLBVH of `in` on the current device.  Returns 0, -1 (bad arguments / device failure, `err` says which), -2 (one of the reference's HLBVH assertions fires on this input).
int build_hlbvh_device(const BuildInput& in, int max_prims_in_node, hipStream_t stream, BuildOutput& out, std::string& err) {
    out = BuildOutput();
    if (in.items || in.n_tris >= 0x3FFFFFFFu) { err = "device build: instanced scenes and more than 2^30 triangles take the host builder"; return -1; }
    const uint32_t n = (uint32_t)in.n_tris;
    if (n == 0) return 0;
    const uint32_t max_prims = (uint32_t)(max_prims_in_node & 0xff);   // bvh/mod.rs:357 `as u8`
    auto t0 = std::chrono::steady_clock::now();
    std::vector<void*> allocs;
    auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); return nullptr; } allocs.push_back(p); return p; };
    size_t n_verts = 0;
    for (size_t i = 0; i < 3 * (size_t)n; i++) n_verts = std::max<size_t>(n_verts, (size_t)in.idx[i] + 1);
    float* dP = (float*)dalloc(n_verts * 12); uint32_t* dIdx = (uint32_t*)dalloc((size_t)n * 12);
    float* blo = (float*)dalloc((size_t)n * 12); float* bhi = (float*)dalloc((size_t)n * 12);
    uint32_t* gb = (uint32_t*)dalloc(32);
    uint32_t* keys[2] = {(uint32_t*)dalloc((size_t)n * 4), (uint32_t*)dalloc((size_t)n * 4)};
    uint32_t* vals[2] = {(uint32_t*)dalloc((size_t)n * 4), (uint32_t*)dalloc((size_t)n * 4)};
    const uint32_t nb = std::min<uint32_t>(1024u, (n + 2047u) / 2048u), per = (((n + nb - 1u) / nb) + 255u) & ~255u;
    uint32_t* bh = (uint32_t*)dalloc((size_t)256 * nb * 4);
    uint32_t* d_start = (uint32_t*)dalloc(4097 * 4);
    uint32_t* leaf_last = (uint32_t*)dalloc((size_t)n * 4);
    phd::DNode* pool = (phd::DNode*)dalloc((size_t)2 * n * sizeof(phd::DNode));
    uint32_t* d_flags = in.tri_flags ? (uint32_t*)dalloc((size_t)n * 4) : nullptr; uint32_t* d_mesh = in.tri_mesh ? (uint32_t*)dalloc((size_t)n * 4) : nullptr;
    TriRec* d_tris = (TriRec*)dalloc((size_t)n * sizeof(TriRec));
    uint32_t *d_tl_first = nullptr, *d_tl_n = nullptr, *d_tl_dense = nullptr, *d_tl_out = nullptr, *d_key_out = nullptr; phd::TreeletInfo* d_info = nullptr; Node64* d_nodes = nullptr;
    std::vector<uint32_t> start(4097), tl_first, tl_n, tl_key;
    std::vector<phd::TreeletInfo> info;
    std::vector<UpperNode> upper; int upper_root = 0;
    int rc = -1;
    for (void* p : allocs) if (!p) { err = "device build: out of device memory"; goto fail; }
    {
        static const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
        PHD_CHECK(hipMemcpyAsync(dP, in.P, n_verts * 12, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemcpyAsync(dIdx, in.idx, (size_t)n * 12, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemcpyAsync(gb, init, 32, hipMemcpyHostToDevice, stream));
        if (d_flags) PHD_CHECK(hipMemcpyAsync(d_flags, in.tri_flags, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        if (d_mesh) PHD_CHECK(hipMemcpyAsync(d_mesh, in.tri_mesh, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        const dim3 g((n + 255u) / 256u), b(256);
        hipLaunchKernelGGL(phd::prim_bounds_kernel, g, b, 0, stream, dP, dIdx, n, blo, bhi, gb);
        hipLaunchKernelGGL(phd::morton_kernel, g, b, 0, stream, blo, bhi, n, gb, keys[0], vals[0]);
        int cur = 0;
        for (int pass = 0; pass < 4; pass++) {   // 32 >= 30 code bits
            hipLaunchKernelGGL(phd::rs_hist_kernel, dim3(nb), dim3(PHD_RS_BLOCK), 0, stream, keys[cur], n, per, pass * 8, bh);
            hipLaunchKernelGGL(phd::scan_kernel, dim3(1), dim3(1024), 0, stream, bh, 256u * nb);
            hipLaunchKernelGGL(phd::rs_scatter_kernel, dim3(nb), dim3(PHD_RS_BLOCK), 0, stream, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, per, pass * 8, bh);
            cur ^= 1;
        }
        uint32_t* codes = keys[cur]; uint32_t* ids = vals[cur];
        hipLaunchKernelGGL(phd::treelet_start_kernel, dim3(17), dim3(256), 0, stream, codes, n, d_start);
        PHD_CHECK(hipGetLastError());
        PHD_CHECK(hipMemcpyAsync(start.data(), d_start, 4097 * 4, hipMemcpyDeviceToHost, stream));
        PHD_CHECK(hipStreamSynchronize(stream));
        for (uint32_t k = 0; k < 4096u; k++) if (start[k + 1] > start[k]) { tl_first.push_back(start[k]); tl_n.push_back(start[k + 1] - start[k]); tl_key.push_back(k); }
        const uint32_t nt = (uint32_t)tl_first.size();
        d_tl_first = (uint32_t*)dalloc((size_t)nt * 4); d_tl_n = (uint32_t*)dalloc((size_t)nt * 4); d_tl_dense = (uint32_t*)dalloc((size_t)nt * 4); d_tl_out = (uint32_t*)dalloc((size_t)nt * 4);
        d_key_out = (uint32_t*)dalloc(4096 * 4); d_info = (phd::TreeletInfo*)dalloc((size_t)nt * sizeof(phd::TreeletInfo));
        if (!d_tl_first || !d_tl_n || !d_tl_dense || !d_tl_out || !d_key_out || !d_info) { err = "device build: out of device memory"; goto fail; }
        PHD_CHECK(hipMemcpyAsync(d_tl_first, tl_first.data(), (size_t)nt * 4, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemcpyAsync(d_tl_n, tl_n.data(), (size_t)nt * 4, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemsetAsync(pool, 0xFF, (size_t)2 * n * sizeof(phd::DNode), stream));
        PHD_CHECK(hipMemsetAsync(leaf_last, 0, (size_t)n * 4, stream));
        {   // emit_lbvh, level by level (K5)
            phd::EmitItem* items = (phd::EmitItem*)dalloc((size_t)2 * n * sizeof(phd::EmitItem));
            uint32_t* lvl = (uint32_t*)dalloc((PHD_MAX_LEVELS + 2) * 4); uint32_t* n_items = (uint32_t*)dalloc(16);
            if (!items || !lvl || !n_items) { err = "device build: out of device memory"; goto fail; }
            PHD_CHECK(hipMemsetAsync(lvl, 0, (PHD_MAX_LEVELS + 2) * 4, stream));
            hipLaunchKernelGGL(phd::emit_roots_kernel, dim3((nt + 255u) / 256u), dim3(256), 0, stream, d_tl_first, d_tl_n, nt, items, lvl, n_items, d_info);
            auto grid_of = [&](int L) { const uint64_t most = std::min<uint64_t>((uint64_t)n, (uint64_t)nt << std::min(L, 24)); return dim3((uint32_t)((most + 255u) / 256u)); };   // a level holds at most 2^L nodes per treelet, and never more than n
            for (int L = 0; L < PHD_MAX_LEVELS; L++) {
                hipLaunchKernelGGL(phd::emit_level_kernel, grid_of(L), dim3(256), 0, stream, items, lvl, L, n_items, codes, max_prims, pool, leaf_last, d_info);
                hipLaunchKernelGGL(phd::emit_close_level_kernel, dim3(1), dim3(64), 0, stream, lvl, L, n_items);
            }
            for (int L = PHD_MAX_LEVELS - 1; L >= 0; L--) hipLaunchKernelGGL(phd::emit_up_kernel, grid_of(L), dim3(256), 0, stream, items, lvl, L, ids, blo, bhi, pool);
            for (int L = 0; L < PHD_MAX_LEVELS; L++) hipLaunchKernelGGL(phd::emit_down_kernel, grid_of(L), dim3(256), 0, stream, items, lvl, L, pool, d_info);
        }
        PHD_CHECK(hipGetLastError());
        info.resize(nt);
        PHD_CHECK(hipMemcpyAsync(info.data(), d_info, (size_t)nt * sizeof(phd::TreeletInfo), hipMemcpyDeviceToHost, stream));
        // the treelet roots' bounds for the SAH over them (hlbvh.rs:86-95)
        std::vector<phd::DNode> roots(nt);
        for (uint32_t t = 0; t < nt; t++) PHD_CHECK(hipMemcpyAsync(&roots[t], pool + 2 * (size_t)tl_first[t], sizeof(phd::DNode), hipMemcpyDeviceToHost, stream));
        PHD_CHECK(hipStreamSynchronize(stream));
        std::vector<float> rb(6 * (size_t)nt);
        for (uint32_t t = 0; t < nt; t++) for (int k = 0; k < 3; k++) { rb[6 * (size_t)t + k] = roots[t].lo[k]; rb[6 * (size_t)t + 3 + k] = roots[t].hi[k]; }
        if (build_upper_sah(rb.data(), nt, upper, upper_root) != 0) { rc = -2; err = "the reference's HLBVH build asserts on this input (hlbvh.rs:338/356/418)"; goto fail; }
        // final numbering: Node64 [0, n_upper) = the SAH nodes over the treelets (pre-order), then every treelet's interior nodes; leaves in the depth-first order of the whole tree
        const uint32_t n_upper = (uint32_t)upper.size();
        std::vector<uint32_t> dense_base(nt), out_base(nt), key_out(4096, 0u);
        { uint32_t acc = n_upper; for (uint32_t t = 0; t < nt; t++) { dense_base[t] = acc; acc += info[t].interior; } out.interior_nodes = acc; }
        {   // depth-first walk of the upper tree: the order in which the treelets' leaf ranges follow each other
            uint32_t acc = 0;
            std::vector<int> stack{upper_root};
            while (!stack.empty()) {
                const int v = stack.back(); stack.pop_back();
                if (v < 0) { const uint32_t t = (uint32_t)(-1 - v); out_base[t] = acc; acc += tl_n[t]; }
                else { stack.push_back(upper[(size_t)v].kid[1]); stack.push_back(upper[(size_t)v].kid[0]); }
            }
        }
        for (uint32_t t = 0; t < nt; t++) key_out[tl_key[t]] = out_base[t];
        d_nodes = (Node64*)dalloc(std::max<size_t>(out.interior_nodes, 1) * sizeof(Node64));
        if (!d_nodes) { err = "device build: out of device memory"; goto fail; }
        PHD_CHECK(hipMemcpyAsync(d_tl_dense, dense_base.data(), (size_t)nt * 4, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemcpyAsync(d_tl_out, out_base.data(), (size_t)nt * 4, hipMemcpyHostToDevice, stream));
        PHD_CHECK(hipMemcpyAsync(d_key_out, key_out.data(), 4096 * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(phd::convert_kernel, dim3(8, nt), dim3(256), 0, stream, pool, d_tl_first, d_tl_n, d_tl_dense, d_tl_out, nt, d_nodes);
        hipLaunchKernelGGL(phd::gather_kernel, g, b, 0, stream, ids, leaf_last, codes, d_start, d_key_out, n, dP, dIdx, d_flags, d_mesh, d_tris);
        PHD_CHECK(hipGetLastError());
        // the upper nodes, on the host: a child is another upper node or a treelet root (an interior node of that treelet, or its single leaf)
        auto ref_of = [&](int v) -> uint32_t {
            if (v >= 0) return (uint32_t)v;
            const uint32_t t = (uint32_t)(-1 - v);
            return info[t].interior ? dense_base[t] /* the root is the first interior node its treelet made */ : (PH_LEAF_BIT | out_base[t]);
        };
        auto box_of = [&](int v, float lo[3], float hi[3]) {
            if (v >= 0) { for (int k = 0; k < 3; k++) { lo[k] = upper[(size_t)v].lo[k]; hi[k] = upper[(size_t)v].hi[k]; } }
            else { const uint32_t t = (uint32_t)(-1 - v); for (int k = 0; k < 3; k++) { lo[k] = rb[6 * (size_t)t + k]; hi[k] = rb[6 * (size_t)t + 3 + k]; } }
        };
        out.nodes.resize(out.interior_nodes);
        out.tris.resize(n);
        if (out.interior_nodes) PHD_CHECK(hipMemcpyAsync(out.nodes.data(), d_nodes, out.interior_nodes * sizeof(Node64), hipMemcpyDeviceToHost, stream));
        PHD_CHECK(hipMemcpyAsync(out.tris.data(), d_tris, (size_t)n * sizeof(TriRec), hipMemcpyDeviceToHost, stream));
        PHD_CHECK(hipStreamSynchronize(stream));
        for (uint32_t v = 0; v < n_upper; v++) {
            const UpperNode& u = upper[v];
            Node64& d = out.nodes[v];
            float l0[3], h0[3], l1[3], h1[3];
            box_of(u.kid[0], l0, h0); box_of(u.kid[1], l1, h1);
            d.x0[0] = l0[0]; d.x0[1] = h0[0]; d.y0[0] = l0[1]; d.y0[1] = h0[1]; d.z0[0] = l0[2]; d.z0[1] = h0[2];
            d.x1[0] = l1[0]; d.x1[1] = h1[0]; d.y1[0] = l1[1]; d.y1[1] = h1[1]; d.z1[0] = l1[2]; d.z1[1] = h1[2];
            d.c0 = ref_of(u.kid[0]); d.c1 = ref_of(u.kid[1]); d.axis = (uint32_t)u.axis; d.pad = 0;
        }
        out.root_ref = ref_of(upper_root);
        float rl[3], rh[3];
        box_of(upper_root, rl, rh);
        for (int k = 0; k < 3; k++) { out.root_lo[k] = rl[k]; out.root_hi[k] = rh[k]; }
        int upper_depth = 0;
        { std::vector<std::pair<int, int>> stk{{upper_root, 1}}; while (!stk.empty()) { auto [v, dpt] = stk.back(); stk.pop_back(); if (v >= 0) { upper_depth = std::max(upper_depth, dpt); stk.push_back({upper[(size_t)v].kid[0], dpt + 1}); stk.push_back({upper[(size_t)v].kid[1], dpt + 1}); } } }
        for (uint32_t t = 0; t < nt; t++) { out.leaf_nodes += info[t].leaves; out.max_leaf_prims = std::max<size_t>(out.max_leaf_prims, info[t].max_leaf); out.max_depth = std::max(out.max_depth, upper_depth + (int)info[t].depth); }
        out.total_nodes = out.interior_nodes + out.leaf_nodes;
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rc = 0;
    }
fail:
    for (void* p : allocs) if (p) (void)hipFree(p);
    return rc;
}

}  // namespace phost

// Test / measurement hook with the signature of pbrt_hip_host_build_bvh (host_setup.cpp): the same tree, built on device `device`.
extern "C" int pbrt_hip_device_build_bvh(int device, const float* P, const uint32_t* idx, uint64_t n_tris, int split_method, int max_prims_in_node, uint32_t* out_ordered_prims,
                                         uint32_t* out_leaf_last, void* out_nodes, uint64_t* out_info, float* out_root_bounds, double* out_seconds) {
    return phost::ph_guard(nullptr, "pbrt_hip_device_build_bvh", [&]() -> int {
    if (split_method != 0 && split_method != 1) return -1;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    phost::BuildInput in{P, idx, (size_t)n_tris, nullptr, nullptr};
    phost::BuildOutput out;
    std::string err;
    const int rc = split_method == 1 ? phost::build_hlbvh_device(in, max_prims_in_node, nullptr, out, err) : phost::build_sah_device(in, max_prims_in_node, nullptr, out, err);
    if (rc && !err.empty()) std::fprintf(stderr, "pbrt_hip_device_build_bvh: %s\n", err.c_str());
    if (rc) return rc;
    for (size_t i = 0; i < out.tris.size(); i++) {
        if (out_ordered_prims) out_ordered_prims[i] = out.tris[i].prim;
        if (out_leaf_last) out_leaf_last[i] = (out.tris[i].flags & PH_TRI_LAST) ? 1u : 0u;
    }
    if (out_nodes && !out.nodes.empty()) std::memcpy(out_nodes, out.nodes.data(), out.nodes.size() * sizeof(Node64));
    if (out_info) { out_info[0] = out.interior_nodes; out_info[1] = out.leaf_nodes; out_info[2] = out.max_leaf_prims; out_info[3] = (uint64_t)out.max_depth; out_info[4] = out.root_ref; }
    if (out_root_bounds) { for (int k = 0; k < 3; k++) { out_root_bounds[k] = out.root_lo[k]; out_root_bounds[3 + k] = out.root_hi[k]; } }
    if (out_seconds) *out_seconds = out.build_seconds;
    return 0;
    });
}
