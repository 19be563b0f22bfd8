// Shade-side work queues (round 4): every wavefront round's live paths regrouped by WHAT THE SHADE SIDE HAS TO DO FOR THEM.
//
// The reference shades one path at a time (sampler_integrator.rs:352-398 -> path.rs:116-279): which thread handles which path, and in which order, is a free
// choice here — every path owns its state record and its film sample.  Rounds 1 - 3 walked the live list in queue order, so a wave held whatever materials
// its 64 paths happened to hit (28 on configs[4]): the texture pass ran a quarter full (`continue` past paths with nothing to evaluate), the general-BSDF
// shade pass a third full (a 256-path block-local sort left ~9 paths per class).  These kernels produce `order`, a permutation of the round's list positions
// grouped by a per-material key the host assigns (materials with per-hit textures or a bump map FIRST, so that the texture pass's work is the prefix
// order[0 .. bin_start[tex_keys])), then the paths whose new vertex only emits (bounce limit reached), then the paths with no new vertex at all (a miss, or only
// pending light-sample results).  The texture and shade passes walk the list through `order`: full waves of ONE material, whose lobe list and texture
// programs are then wave-uniform.
//
// A counting sort shaped like raysort.h's: block-local tallies in LDS -> one global atomic per block and non-empty bin -> a one-block scan -> the same
// tallies again to claim the block's share of each bin.  With a few dozen keys plain per-lane LDS atomics would pile onto the same addresses, so lanes that
// hold the same key are found with a ballot loop and ONE lane adds the wave's count (a wave of coherent paths holds a handful of keys).
#pragma once
#include "traverse.h"

namespace ph {

#define PH_MS_BINS 1024u           // keys in use: < PH_MS_BINS (materials beyond the bins share the last textured / untextured key: still correct, only less sorted)
#define PH_MS_BLOCK 256

struct MatSortParams {
    const uint4* s_idx;            // the round's path records (ext slot in .x, flags | bounces << 8 in .w)
    const HitOut* hits;
    const uint32_t* n_live;
    int32_t max_depth;
    const uint16_t* mat_key;       // per material: its key (host: upload_scene)
    uint32_t key_emit, key_idle;   // the two keys behind the materials'
    const TriRec* tris; const MeshRec* meshes;   // only for a hit whose TriRec could not carry its material id (more than 4094 materials)
    uint16_t* keys;                // out: key per list position (hist pass), read again by the scatter pass
    uint32_t* order;               // out: list positions grouped by key
    uint32_t* bin_start;           // [PH_MS_BINS + 1] tallies -> exclusive starts; [key_idle + 1] = n_live
    uint32_t* bin_cursor;          // [PH_MS_BINS]
};

PH_DEV void matsort_slice(uint32_t n, uint32_t& lo, uint32_t& hi) {
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + PH_MS_BLOCK - 1u) & ~(uint32_t)(PH_MS_BLOCK - 1u);
    lo = blockIdx.x * per; hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}

// what the shade side has to do for the path at list position i
PH_DEV uint32_t matsort_key_of(const MatSortParams& p, uint32_t i) {
    const uint4 c4 = p.s_idx[i];
    const uint32_t flags = c4.w & 0xffu, bounces = (c4.w >> 8) & 0xffu;
    if (!(flags & 1u /* F_EXT */)) return p.key_idle;
    const float4* hp = reinterpret_cast<const float4*>(p.hits + c4.x);
    const float4 h0 = hp[0];
    if (__float_as_uint(h0.y) == 0xFFFFFFFFu) return p.key_idle;
    if ((int)bounces >= p.max_depth) return p.key_emit;
    const float4 h1 = hp[1];
    uint32_t mat = (__float_as_uint(h1.w) >> 3) & 0xFFFu;   // HitOut::pad[2] = class | material << 3 (traverse.h: from the TriRec's flags)
    if (mat == 0xFFFu) mat = p.meshes[p.tris[__float_as_uint(h1.y)].mesh].material;
    return p.mat_key[mat];
}

__global__ __launch_bounds__(PH_MS_BLOCK) void matsort_hist_kernel(MatSortParams p) {
    __shared__ uint32_t h[PH_MS_BINS];
    const uint32_t n = *p.n_live;
    uint32_t lo, hi;
    matsort_slice(n, lo, hi);
    if (lo >= hi) return;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = threadIdx.x; k < PH_MS_BINS; k += PH_MS_BLOCK) h[k] = 0u;
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += PH_MS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < hi;
        uint32_t key = 0u;
        if (valid) { key = matsort_key_of(p, i); p.keys[i] = (uint16_t)key; }
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader);
            const uint64_t m = __ballot(valid && key == k);
            if ((int)lane == leader) atomicAdd(&h[k], (uint32_t)__popcll(m));
            todo &= ~m;
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PH_MS_BINS; k += PH_MS_BLOCK)
        if (h[k]) atomicAdd(&p.bin_start[k], h[k]);
}

// one block: exclusive scan of the tallies in place (+ the total behind them), cursors cleared
__global__ __launch_bounds__(PH_MS_BINS) void matsort_scan_kernel(MatSortParams p) {
    __shared__ uint32_t part[PH_MS_BINS];
    const uint32_t v = p.bin_start[threadIdx.x];
    part[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < PH_MS_BINS; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    p.bin_start[threadIdx.x] = part[threadIdx.x] - v;
    p.bin_cursor[threadIdx.x] = 0u;
    if (threadIdx.x == PH_MS_BINS - 1u) p.bin_start[PH_MS_BINS] = part[threadIdx.x];
}

__global__ __launch_bounds__(PH_MS_BLOCK) void matsort_scatter_kernel(MatSortParams p) {
    __shared__ uint32_t h[PH_MS_BINS];     // tallies of the slice, then the running rank inside each bin
    __shared__ uint32_t base_of[PH_MS_BINS];  // where the slice's share of each bin starts in `order`
    const uint32_t n = *p.n_live;
    uint32_t lo, hi;
    matsort_slice(n, lo, hi);
    if (lo >= hi) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t k = threadIdx.x; k < PH_MS_BINS; k += PH_MS_BLOCK) h[k] = 0u;
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += PH_MS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < hi;
        const uint32_t key = valid ? (uint32_t)p.keys[i] : 0u;
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader);
            const uint64_t m = __ballot(valid && key == k);
            if ((int)lane == leader) atomicAdd(&h[k], (uint32_t)__popcll(m));
            todo &= ~m;
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PH_MS_BINS; k += PH_MS_BLOCK) {
        const uint32_t c = h[k];
        base_of[k] = c ? p.bin_start[k] + atomicAdd(&p.bin_cursor[k], c) : 0u;
        h[k] = 0u;
    }
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += PH_MS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < hi;
        const uint32_t key = valid ? (uint32_t)p.keys[i] : 0u;
        uint64_t todo = __ballot(valid);
        uint32_t slot = 0u;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader);
            const uint64_t m = __ballot(valid && key == k);
            uint32_t wave_base = 0u;
            if ((int)lane == leader) wave_base = atomicAdd(&h[k], (uint32_t)__popcll(m));
            wave_base = (uint32_t)__shfl((int)wave_base, leader);
            if (valid && key == k) slot = base_of[k] + wave_base + (uint32_t)__popcll(m & lane_lt);
            todo &= ~m;
        }
        if (valid) p.order[slot] = i;   // (a slice's share of a bin keeps the slice's order: neighbours in the list stay neighbours in the bin)
    }
}

}  // namespace ph
