// Ray binning between wavefront rounds ("ray sorting / Morton ordering").
//
// The reference traces one path at a time (sampler_integrator.rs:352-398), so the order in which a round's rays are traced is a free
// choice here: every ray's result goes to its own slot and nothing else depends on who traces it when.  After the first diffuse bounce
// the queues are in path order, i.e. spatially random, and on scenes larger than the caches every node fetch of a ray then misses L2.
// These kernels produce `order`, a permutation of the round's queue positions grouped by the cell of the ray origin (Morton order
// of a 16^3 grid over the scene bound, optionally 8^3 cells x direction octant), closest-hit rays first, any-hit rays behind them —
// the traversal kernel walks the queue through it.  Rays are not moved: the kernel that writes a ray (shade) also writes its 4-byte bin key, the two
// passes here read the keys and write 4 B of `order` per ray.
//
// A counting sort without global atomics per ray: every block histograms its contiguous slice of the queue in LDS and adds the
// non-empty bins to the global tallies (one atomic per block and bin), a single block scans the 8192 tallies, and the second
// pass repeats the LDS histogram, claims the block's share of each bin with one atomic per bin and ranks its rays inside LDS.
// The order inside a bin is whatever the atomics give; it does not matter.  (Tried and dropped: one LDS atomic per distinct key of a wave, found with a
// ballot loop — a wave of the path-ordered queue holds many distinct keys, the loop cost more than the plain atomics: binning 28 -> 70 ms per configs[2] frame.)
#pragma once
#include "traverse.h"

namespace ph {

#define PH_SORT_BINS 4096u            // per ray kind
#define PH_SORT_KEYS (2u * PH_SORT_BINS)
#define PH_SORT_BLOCK 256

struct RaySortParams {
    const uint32_t* keys_cl; const uint32_t* keys_sh;   // one key per queue slot, written by the kernel that wrote the ray (ray_sort_key)
    const uint32_t* n_cl; const uint32_t* n_sh;
    uint32_t* order;        // out: [n_cl + n_sh]
    uint32_t* bin_start;    // [PH_SORT_KEYS] tallies -> exclusive starts (scan kernel)
    uint32_t* bin_cursor;   // [PH_SORT_KEYS] zeroed by the scan kernel
};
struct RaySortGrid {        // how a ray is binned
    float lo[3], scale[3];  // cell coordinate = (o - lo) * scale in [0, 1)
    uint32_t mode;          // 0: off; 1: 16^3 origin cells; 2: 8^3 origin cells x direction octant
};

PH_DEV uint32_t part1by2(uint32_t v) {  // 4 bits -> every third bit
    v &= 0xFu;
    v = (v | (v << 4)) & 0xC3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}

// bin of a ray with origin (ox, oy, oz) and direction (dx, dy, dz); < PH_SORT_BINS
PH_DEV uint32_t ray_sort_key(const RaySortGrid& g, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float cells = (g.mode == 2u || g.mode == 3u) ? 8.0f : 16.0f;   // (mode 3: 8^3 cells without the octant — a measurement aid that isolates what the octant is worth)
    const float fx = (ox - g.lo[0]) * g.scale[0], fy = (oy - g.lo[1]) * g.scale[1], fz = (oz - g.lo[2]) * g.scale[2];
    // NaN / out-of-bound origins land in the border cells: any bin is a correct bin
    const uint32_t cx = (uint32_t)pmini(pmaxi((int)(fx * cells), 0), (int)cells - 1);
    const uint32_t cy = (uint32_t)pmini(pmaxi((int)(fy * cells), 0), (int)cells - 1);
    const uint32_t cz = (uint32_t)pmini(pmaxi((int)(fz * cells), 0), (int)cells - 1);
    uint32_t key = part1by2(cx) | (part1by2(cy) << 1) | (part1by2(cz) << 2);
    if (g.mode == 2u) key = (key << 3) | (dx < 0.0f ? 1u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 4u : 0u);
    return key & (PH_SORT_BINS - 1u);
}
PH_DEV uint32_t raysort_key_at(const RaySortParams& p, uint32_t i, uint32_t n_cl) {
    return i < n_cl ? p.keys_cl[i] : PH_SORT_BINS + p.keys_sh[i - n_cl];
}

PH_DEV void raysort_slice(const RaySortParams& p, uint32_t n, uint32_t& lo, uint32_t& hi) {
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + PH_SORT_BLOCK - 1u) & ~(uint32_t)(PH_SORT_BLOCK - 1u);
    lo = blockIdx.x * per; hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}

__global__ __launch_bounds__(PH_SORT_BLOCK) void raysort_hist_kernel(RaySortParams p) {
    __shared__ uint32_t h[PH_SORT_KEYS];
    const uint32_t n_cl = *p.n_cl, n = n_cl + *p.n_sh;
    uint32_t lo, hi;
    raysort_slice(p, n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t k = threadIdx.x; k < PH_SORT_KEYS; k += PH_SORT_BLOCK) h[k] = 0u;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PH_SORT_BLOCK) atomicAdd(&h[raysort_key_at(p, i, n_cl)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PH_SORT_KEYS; k += PH_SORT_BLOCK)
        if (h[k]) atomicAdd(&p.bin_start[k], h[k]);
}

// one block: exclusive scan of the PH_SORT_KEYS tallies in place, cursors cleared
__global__ __launch_bounds__(1024) void raysort_scan_kernel(RaySortParams p) {
    __shared__ uint32_t part[1024];
    constexpr uint32_t PER = PH_SORT_KEYS / 1024u;
    uint32_t v[PER], sum = 0u;
    for (uint32_t k = 0; k < PER; k++) { v[k] = p.bin_start[threadIdx.x * PER + k]; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < PER; k++) { p.bin_start[threadIdx.x * PER + k] = run; p.bin_cursor[threadIdx.x * PER + k] = 0u; run += v[k]; }
}

__global__ __launch_bounds__(PH_SORT_BLOCK) void raysort_scatter_kernel(RaySortParams p) {
    __shared__ uint32_t h[PH_SORT_KEYS];     // tallies of the slice, then the running rank inside each bin
    __shared__ uint32_t base[PH_SORT_KEYS];  // where the slice's share of each bin starts in `order`
    const uint32_t n_cl = *p.n_cl, n = n_cl + *p.n_sh;
    uint32_t lo, hi;
    raysort_slice(p, n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t k = threadIdx.x; k < PH_SORT_KEYS; k += PH_SORT_BLOCK) h[k] = 0u;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PH_SORT_BLOCK) atomicAdd(&h[raysort_key_at(p, i, n_cl)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PH_SORT_KEYS; k += PH_SORT_BLOCK) {
        const uint32_t c = h[k];
        base[k] = c ? p.bin_start[k] + atomicAdd(&p.bin_cursor[k], c) : 0u;
        h[k] = 0u;
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PH_SORT_BLOCK) {
        const uint32_t key = raysort_key_at(p, i, n_cl);
        p.order[base[key] + atomicAdd(&h[key], 1u)] = i;
    }
}

}  // namespace ph
