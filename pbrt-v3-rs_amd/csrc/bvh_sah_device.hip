// The reference's default BVH (SplitMethod::SAH, accelerators/src/bvh/mod.rs:43-153, sah.rs:26-367) built on the device, one level of the tree per round of
// kernels.  bvh_sah_steps.h holds what one work item of each pass does and why the result is the reference's tree; this file is the grids around those
// steps, the two reductions that need care on a GPU, and the host loop.
//
//   per level:  decide (per node)  ->  buckets (per primitive; a block whose 2048 positions all belong to one node accumulates in LDS and merges once:
//               the top of the tree would otherwise send every primitive's 13 atomics to the same 156 words)  ->  eval (per node: costs, split, children)
//               ->  predicate + prefix sum (three kernels)  ->  misplaced lists  ->  swaps  ->  relabel
//   afterwards: subtree sizes bottom-up, Node64 numbering top-down (the host builder's numbering, so the arrays are the same arrays), TriRecs.
// Two host round trips per level (the number of bucket slots, the number of nodes made).
#include "bvh_sah_steps.h"
#include "scene_host.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace phs {

// positions [p0, p1): a block whose chunk lies in one tree gathers the tree's bounds in LDS and merges them once, a chunk that straddles trees sends every element's to memory
__global__ __launch_bounds__(256) void k_init(Ctx c, uint32_t p0, uint32_t p1) {
    __shared__ uint32_t sm[12];
    const uint32_t base = p0 + blockIdx.x * PHS_CHUNK, lim = base + PHS_CHUNK < p1 ? base + PHS_CHUNK : p1;
    if (base >= lim) return;
    const uint32_t ta = tree_of(c, base), tb = tree_of(c, lim - 1u);
    if (ta == tb) {
        if (threadIdx.x == 0) init_bounds_words(sm);
        __syncthreads();
        for (uint32_t i = base + threadIdx.x; i < lim; i += 256u) init_elem(c, i, ta, sm);
        __syncthreads();
        if (threadIdx.x < 12u) {
            const uint32_t k = threadIdx.x;
            uint32_t* dst = c.root_words + 12u * ta;
            if (k < 3u || (k >= 6u && k < 9u)) a_min(dst + k, sm[k]); else a_max(dst + k, sm[k]);
        }
    } else {
        for (uint32_t i = base + threadIdx.x; i < lim; i += 256u) { const uint32_t t = tree_of(c, i); init_elem(c, i, t, c.root_words + 12u * t); }
    }
}
__global__ __launch_bounds__(256) void k_root(Ctx c) { const uint32_t t = blockIdx.x * 256u + threadIdx.x; if (t < c.n_trees) make_root(c, t); }
__global__ __launch_bounds__(256) void k_root_numbers(Ctx c, const uint32_t* bases) { const uint32_t t = blockIdx.x * 256u + threadIdx.x; if (t < c.n_trees) root_numbers(c, t, bases[t]); }

__global__ __launch_bounds__(256) void k_decide(Ctx c, uint32_t v0, uint32_t v1) {
    const uint32_t v = v0 + blockIdx.x * 256u + threadIdx.x;
    if (v < v1) decide_node(c, v);
}
__global__ __launch_bounds__(256) void k_scratch_init(Ctx c, uint32_t slots) {
    const size_t words = (size_t)slots * PHS_SLOT;
    for (size_t w = (size_t)blockIdx.x * 256u + threadIdx.x; w < words; w += (size_t)gridDim.x * 256u) c.scratch[w] = scratch_init_word((uint32_t)(w % PHS_SLOT));
}
__global__ __launch_bounds__(256) void k_bucket(Ctx c) {
    __shared__ uint32_t sm[PHS_SLOT];
    const uint32_t base = blockIdx.x * PHS_CHUNK, lim = base + PHS_CHUNK < c.n ? base + PHS_CHUNK : c.n;
    const uint32_t va = c.seg[base], vb = c.seg[lim - 1u];
    // both ends in the same live node: every position between them is in that node's range too (a range is contiguous)
    const bool one_node = va == vb && va != PHS_NONE;
    if (one_node) {
        const uint32_t slot = c.nodes[va].slot;
        if (slot == PHS_NONE) return;
        for (uint32_t w = threadIdx.x; w < (uint32_t)PHS_SLOT; w += 256u) sm[w] = scratch_init_word(w);
        __syncthreads();
        for (uint32_t i = base + threadIdx.x; i < lim; i += 256u) bucket_elem(c, i, sm);
        __syncthreads();
        uint32_t* dst = c.scratch + (size_t)slot * PHS_SLOT;
        for (uint32_t w = threadIdx.x; w < (uint32_t)PHS_SLOT; w += 256u) scratch_merge_word(dst, w, sm[w]);
    } else {
        for (uint32_t i = base + threadIdx.x; i < lim; i += 256u) bucket_elem(c, i, nullptr);
    }
}
__global__ __launch_bounds__(256) void k_eval(Ctx c, uint32_t v0, uint32_t v1) {
    const uint32_t v = v0 + blockIdx.x * 256u + threadIdx.x;
    if (v < v1) eval_node(c, v);
}

// prefix sum of the predicate: (1) flags + per-chunk sums, (2) the chunk sums scanned by one block, (3) the flags replaced by their exclusive prefix
__global__ __launch_bounds__(256) void k_flags(Ctx c, uint32_t* chunk_sums) {
    __shared__ uint32_t part[4];
    const uint32_t base = blockIdx.x * PHS_CHUNK, lim = base + PHS_CHUNK < c.n ? base + PHS_CHUNK : c.n;
    uint32_t sum = 0u;
    for (uint32_t i = base + threadIdx.x; i < lim; i += 256u) { const uint32_t f = flag_of(c, i); c.scan[i] = f; sum += f; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void k_scan_chunks(uint32_t* a, uint32_t m) {   // exclusive scan of m counters in place, one block
    __shared__ uint32_t part[1024];
    const uint32_t per = (m + 1023u) / 1024u, lo = threadIdx.x * per < m ? threadIdx.x * per : m, hi = lo + per < m ? lo + per : m;
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
__global__ __launch_bounds__(256) void k_scan_write(Ctx c, const uint32_t* chunk_base) {
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * PHS_CHUNK;
    const uint32_t per = PHS_CHUNK / 256u, first = base + threadIdx.x * per;
    uint32_t f[PHS_CHUNK / 256u];
    uint32_t sum = 0u;
    for (uint32_t j = 0; j < per; j++) { f[j] = first + j < c.n ? c.scan[first + j] : 0u; sum += f[j]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = chunk_base[blockIdx.x] + part[threadIdx.x] - sum;
    for (uint32_t j = 0; j < per; j++) if (first + j < c.n) { c.scan[first + j] = run; run += f[j]; }
}

__global__ __launch_bounds__(256) void k_lists(Ctx c) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < c.n) list_elem(c, i); }
__global__ __launch_bounds__(256) void k_swap(Ctx c) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < c.n) swap_pos(c, i); }
__global__ __launch_bounds__(256) void k_relabel(Ctx c) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < c.n) relabel_pos(c, i); }
__global__ __launch_bounds__(256) void k_size(Ctx c, uint32_t v0, uint32_t v1) { const uint32_t v = v0 + blockIdx.x * 256u + threadIdx.x; if (v < v1) size_node(c, v); }
__global__ __launch_bounds__(256) void k_number(Ctx c, uint32_t v0, uint32_t v1) { const uint32_t v = v0 + blockIdx.x * 256u + threadIdx.x; if (v < v1) number_node(c, v); }
__global__ __launch_bounds__(256) void k_emit(Ctx c) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < c.n) emit_tri(c, i); }

}  // namespace phs

namespace phost {

#define PHS_CHECK(call)                                                                 \
    do { hipError_t e__ = (call); if (e__ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e__); (void)hipGetLastError(); goto fail; } } while (0)

// Builds the SAH tree of `in` on the current device — or, with `forest`, the scene's aggregate and the aggregates of its instanced objects in one go (bvh_sah_steps.h: Ctx).
// Returns 0, or -1 (bad arguments / device failure, `err` says which).
int build_sah_device(const BuildInput& in, int max_prims_in_node, hipStream_t stream, BuildOutput& out, std::string& err, void** keep_nodes, void** keep_tris,
                     const ForestSpec* forest, std::vector<ForestTreeOut>* trees_out) {
    using namespace phs;
    out = BuildOutput();
    if (trees_out) trees_out->clear();
    if ((in.items && !forest) || in.n_tris >= 0x3FFFFFFFu || (forest && (forest->n_items >= 0x3FFFFFFFu || forest->n_trees == 0 || !forest->tree_start || !forest->items))) {
        err = "device build: bad arguments (an item list needs a forest description; at most 2^30 primitives)"; return -1;
    }
    const uint32_t n = (uint32_t)(forest ? forest->n_items : in.n_tris);
    if (n == 0) return 0;
    const uint32_t n_trees = forest ? forest->n_trees : 1u;
    const uint32_t one_tree[2] = {0u, n};
    const uint32_t* tree_start = forest ? forest->tree_start : one_tree;
    for (uint32_t t = 0; t < n_trees; t++) if (tree_start[t] >= tree_start[t + 1] || tree_start[t + 1] > n) { err = "device build: empty or unordered tree range"; return -1; }
    if (tree_start[0] != 0 || tree_start[n_trees] != n) { err = "device build: tree ranges do not cover the items"; return -1; }
    auto t0 = std::chrono::steady_clock::now();
    const bool prof = std::getenv("PBRT_HIP_BUILD_PROFILE") != nullptr;
    auto lap = [&](const char* what) { if (prof) { (void)hipStreamSynchronize(stream); std::fprintf(stderr, "build_sah_device n=%u: %-28s at %.3f s\n", n, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } };
    std::vector<void*> allocs;
    auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); return nullptr; } allocs.push_back(p); return p; };
    size_t n_verts = 0;
    for (size_t i = 0; i < 3 * in.n_tris; i++) n_verts = std::max<size_t>(n_verts, (size_t)in.idx[i] + 1);
    const uint32_t n_chunks = (n + PHS_CHUNK - 1u) / PHS_CHUNK;
    const size_t n_inst = forest ? forest->n_inst : 0;
    float* dP = (float*)dalloc(n_verts * 12); uint32_t* dIdx = (uint32_t*)dalloc(in.n_tris * 12);
    uint32_t* d_flags = in.tri_flags ? (uint32_t*)dalloc(in.n_tris * 4) : nullptr; uint32_t* d_mesh = in.tri_mesh ? (uint32_t*)dalloc(in.n_tris * 4) : nullptr;
    uint32_t* d_items = forest ? (uint32_t*)dalloc((size_t)n * 4) : nullptr;
    float* d_inst_bounds = n_inst ? (float*)dalloc(n_inst * 24) : nullptr;
    uint32_t* d_tree_start = (uint32_t*)dalloc(((size_t)n_trees + 1) * 4);
    uint32_t* d_root_words = (uint32_t*)dalloc((size_t)n_trees * 48);
    uint32_t* d_bases = (uint32_t*)dalloc((size_t)n_trees * 4);
    Ctx c{};
    c.n = n; c.max_prims = (uint32_t)(max_prims_in_node & 0xff);   // bvh/mod.rs:357 `as u8`
    c.P = dP; c.idx = dIdx; c.tri_flags = d_flags; c.tri_mesh = d_mesh;
    c.items = d_items; c.inst_bounds = d_inst_bounds; c.tree_start = d_tree_start; c.n_trees = n_trees; c.root_words = d_root_words;
    c.e_lo = (Elem*)dalloc((size_t)n * sizeof(Elem)); c.e_hi = (Elem*)dalloc((size_t)n * sizeof(Elem));
    c.seg = (uint32_t*)dalloc((size_t)n * 4); c.bkt = (uint8_t*)dalloc(n);
    c.nodes = (SNode*)dalloc(((size_t)2 * n + 2) * sizeof(SNode));
    c.counters = (uint32_t*)dalloc(64);
    c.scan = (uint32_t*)dalloc(((size_t)n + 1) * 4); c.lf = (uint32_t*)dalloc((size_t)n * 4); c.lb = (uint32_t*)dalloc((size_t)n * 4);
    c.leaf_last = (uint32_t*)dalloc((size_t)n * 4);
    c.out_tris = (TriRec*)dalloc((size_t)n * sizeof(TriRec));
    uint32_t* chunk_sums = (uint32_t*)dalloc((size_t)n_chunks * 4);
    c.scratch = nullptr; c.out_nodes = nullptr;
    uint32_t scratch_slots = 0;
    std::vector<std::pair<uint32_t, uint32_t>> levels;
    uint32_t host_counters[16];
    std::vector<uint32_t> root_words((size_t)n_trees * 12), bases(n_trees);
    std::vector<SNode> roots(n_trees);
    std::vector<float> inst_bounds(n_inst * 6);
    int rc = -1;
    for (void* p : allocs) if (!p) { err = "device build: out of device memory"; goto fail; }
    {
        uint32_t init[16] = {0};
        for (uint32_t t = 0; t < n_trees; t++) init_bounds_words(root_words.data() + 12 * (size_t)t);
        PHS_CHECK(hipMemcpyAsync(dP, in.P, n_verts * 12, hipMemcpyHostToDevice, stream));
        PHS_CHECK(hipMemcpyAsync(dIdx, in.idx, in.n_tris * 12, hipMemcpyHostToDevice, stream));
        PHS_CHECK(hipMemcpyAsync(c.counters, init, 64, hipMemcpyHostToDevice, stream));
        PHS_CHECK(hipMemcpyAsync(d_tree_start, tree_start, ((size_t)n_trees + 1) * 4, hipMemcpyHostToDevice, stream));
        PHS_CHECK(hipMemcpyAsync(d_root_words, root_words.data(), (size_t)n_trees * 48, hipMemcpyHostToDevice, stream));
        if (d_items) PHS_CHECK(hipMemcpyAsync(d_items, forest->items, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        if (d_flags) PHS_CHECK(hipMemcpyAsync(d_flags, in.tri_flags, in.n_tris * 4, hipMemcpyHostToDevice, stream));
        if (d_mesh) PHS_CHECK(hipMemcpyAsync(d_mesh, in.tri_mesh, in.n_tris * 4, hipMemcpyHostToDevice, stream));
        PHS_CHECK(hipMemsetAsync(c.leaf_last, 0, (size_t)n * 4, stream));
        const dim3 per_elem((n + 255u) / 256u), per_chunk(n_chunks), b(256);
        lap("allocations and uploads");
        // the objects' primitives first: the scene-level tree holds TransformedPrimitives whose bounds are the objects' root bounds carried to world space
        const uint32_t top_end = n_inst ? tree_start[1] : 0u;
        if (n_inst) {
            if (n_trees < 2) { err = "device build: instances without object trees"; goto fail; }
            hipLaunchKernelGGL(k_init, dim3((n - top_end + PHS_CHUNK - 1u) / PHS_CHUNK), b, 0, stream, c, top_end, n);
            PHS_CHECK(hipMemcpyAsync(root_words.data(), d_root_words, (size_t)n_trees * 48, hipMemcpyDeviceToHost, stream));
            PHS_CHECK(hipStreamSynchronize(stream));
            for (size_t k = 0; k < n_inst; k++) {
                const uint32_t t = forest->inst_tree[k];
                if (t == 0 || t >= n_trees) { err = "device build: an instance names no object tree"; goto fail; }
                float lo[3], hi[3];
                for (int q = 0; q < 3; q++) { lo[q] = ord2f(root_words[12 * (size_t)t + q]); hi[q] = ord2f(root_words[12 * (size_t)t + 3 + q]); }
                transform_bounds(forest->inst_i2w + 16 * k, lo, hi, inst_bounds.data() + 6 * k);
            }
            PHS_CHECK(hipMemcpyAsync(d_inst_bounds, inst_bounds.data(), n_inst * 24, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_init, dim3((top_end + PHS_CHUNK - 1u) / PHS_CHUNK), b, 0, stream, c, 0u, top_end);
        } else {
            hipLaunchKernelGGL(k_init, per_chunk, b, 0, stream, c, 0u, n);
        }
        hipLaunchKernelGGL(k_root, dim3((n_trees + 255u) / 256u), b, 0, stream, c);
        uint32_t lv0 = 0, lv1 = n_trees;
        int depth = 0;
        while (lv0 < lv1) {
            if (levels.size() > 4096) { err = "device build: the tree is deeper than 4096 levels"; goto fail; }
            levels.push_back({lv0, lv1});
            const dim3 per_node((lv1 - lv0 + 255u) / 256u);
            PHS_CHECK(hipMemsetAsync(c.counters + 1, 0, 4, stream));
            hipLaunchKernelGGL(k_decide, per_node, b, 0, stream, c, lv0, lv1);
            PHS_CHECK(hipMemcpyAsync(host_counters, c.counters, 8, hipMemcpyDeviceToHost, stream));
            PHS_CHECK(hipStreamSynchronize(stream));
            const uint32_t slots = host_counters[1];
            if (slots) {
                if (slots > scratch_slots) {   // grow-only; the old block is freed with the rest at the end
                    scratch_slots = std::max(slots, scratch_slots * 2u);
                    c.scratch = (uint32_t*)dalloc((size_t)scratch_slots * PHS_SLOT * 4);
                    if (!c.scratch) { err = "device build: out of device memory"; goto fail; }
                }
                hipLaunchKernelGGL(k_scratch_init, dim3(std::min<uint32_t>(4096u, (uint32_t)(((size_t)slots * PHS_SLOT + 255u) / 256u))), b, 0, stream, c, slots);
                hipLaunchKernelGGL(k_bucket, per_chunk, b, 0, stream, c);
                hipLaunchKernelGGL(k_eval, per_node, b, 0, stream, c, lv0, lv1);
                hipLaunchKernelGGL(k_flags, per_chunk, b, 0, stream, c, chunk_sums);
                hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, stream, chunk_sums, n_chunks);
                hipLaunchKernelGGL(k_scan_write, per_chunk, b, 0, stream, c, chunk_sums);
                hipLaunchKernelGGL(k_lists, per_elem, b, 0, stream, c);
                hipLaunchKernelGGL(k_swap, per_elem, b, 0, stream, c);
            }
            hipLaunchKernelGGL(k_relabel, per_elem, b, 0, stream, c);
            PHS_CHECK(hipGetLastError());
            PHS_CHECK(hipMemcpyAsync(host_counters, c.counters, 4, hipMemcpyDeviceToHost, stream));
            PHS_CHECK(hipStreamSynchronize(stream));
            if (host_counters[0] > lv1) depth = (int)levels.size();
            if (host_counters[0] > 2u * n) { err = "device build: node count out of range"; goto fail; }
            lv0 = lv1; lv1 = host_counters[0];
        }
        lap("tree levels");
        for (size_t L = levels.size(); L-- > 0;) hipLaunchKernelGGL(k_size, dim3((levels[L].second - levels[L].first + 255u) / 256u), b, 0, stream, c, levels[L].first, levels[L].second);
        PHS_CHECK(hipMemcpyAsync(roots.data(), c.nodes, (size_t)n_trees * sizeof(SNode), hipMemcpyDeviceToHost, stream));
        PHS_CHECK(hipStreamSynchronize(stream));
        out.interior_nodes = 0;
        for (uint32_t t = 0; t < n_trees; t++) { bases[t] = (uint32_t)out.interior_nodes; out.interior_nodes += roots[t].size; }
        PHS_CHECK(hipMemcpyAsync(d_bases, bases.data(), (size_t)n_trees * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_root_numbers, dim3((n_trees + 255u) / 256u), b, 0, stream, c, d_bases);
        c.out_nodes = (Node64*)dalloc(std::max<size_t>(out.interior_nodes, 1) * sizeof(Node64));
        if (!c.out_nodes) { err = "device build: out of device memory"; goto fail; }
        for (size_t L = 0; L < levels.size(); L++) hipLaunchKernelGGL(k_number, dim3((levels[L].second - levels[L].first + 255u) / 256u), b, 0, stream, c, levels[L].first, levels[L].second);
        hipLaunchKernelGGL(k_emit, per_elem, b, 0, stream, c);
        PHS_CHECK(hipGetLastError());
        lap("sizes, numbering, records");
        if (keep_nodes && keep_tris) {   // the tree stays where it is
            *keep_nodes = c.out_nodes; *keep_tris = c.out_tris;
            for (void*& p : allocs) if (p == (void*)c.out_nodes || p == (void*)c.out_tris) p = nullptr;
        } else {
            out.nodes.resize(out.interior_nodes);
            out.tris.resize(n);
            lap("host vectors");
            if (out.interior_nodes) PHS_CHECK(hipMemcpyAsync(out.nodes.data(), c.out_nodes, out.interior_nodes * sizeof(Node64), hipMemcpyDeviceToHost, stream));
            PHS_CHECK(hipMemcpyAsync(out.tris.data(), c.out_tris, (size_t)n * sizeof(TriRec), hipMemcpyDeviceToHost, stream));
        }
        PHS_CHECK(hipMemcpyAsync(host_counters, c.counters, 64, hipMemcpyDeviceToHost, stream));
        PHS_CHECK(hipStreamSynchronize(stream));
        lap("downloads");
        auto ref_of = [&](uint32_t t) { return roots[t].size ? bases[t] : (PH_LEAF_BIT | tree_start[t]); };
        out.root_ref = ref_of(0);
        for (int k = 0; k < 3; k++) { out.root_lo[k] = roots[0].lo[k]; out.root_hi[k] = roots[0].hi[k]; }
        if (trees_out)
            for (uint32_t t = 0; t < n_trees; t++) {
                ForestTreeOut fo; fo.root_ref = ref_of(t); fo.n_items = tree_start[t + 1] - tree_start[t];
                for (int k = 0; k < 3; k++) { fo.lo[k] = roots[t].lo[k]; fo.hi[k] = roots[t].hi[k]; }
                trees_out->push_back(fo);
            }
        out.leaf_nodes = host_counters[2]; out.max_leaf_prims = host_counters[3]; out.max_depth = depth;   // over all trees of a forest
        out.total_nodes = out.interior_nodes + out.leaf_nodes;
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rc = 0;
    }
fail:
    for (void* p : allocs) if (p) (void)hipFree(p);
    return rc;
}

}  // namespace phost
