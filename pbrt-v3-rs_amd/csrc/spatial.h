// SpatialLightDistribution (core/src/light_distrib/spatial.rs) for the wavefront renderer.
//
// The reference keeps one Distribution1D per voxel of a <=64^3 grid over the scene bounds in a lock-free hash table,
// filled by whichever thread first looks a voxel up (:163-244).  A voxel's distribution is a pure function of the voxel
// (128 Halton points inside it x every light's sample_li, :90-160), so the table is only a cache.  Here it becomes a
// dense voxel -> slot table plus a slot pool in HBM, filled between the extend and shade stages of an iteration:
//
//   spatial_mark     one thread per live path: voxel of the new vertex; first toucher claims a pool slot  (lookup, :166-181)
//   spatial_compute  one block per newly claimed voxel: light_contrib[j] over 128 samples, thread-parallel over lights;
//                    the reference's left-to-right f32 sums (sum_contrib, the CDF) are kept sequential, staged through LDS
//   shade            reads func / cdf / func_int of the vertex's voxel instead of the scene-wide arrays
//
// (The reference's lookup can hand back None while another thread is still filling the entry, :213-216, and the caller
// then samples uniformly: a data race that exists only with >1 thread.  This is the race-free behaviour.)
#pragma once
#include "pt_device.h"

namespace ph {

struct SpatialRec {
    uint32_t enabled;
    int32_t nv[3];
    float lo[3], hi[3];      // scene.world_bound
    int32_t* vox_slot;       // nv[0]*nv[1]*nv[2] entries: -1 empty, -2 claimed but not computable (pool exhausted), >= 0 slot
    float* pool;             // capacity * stride floats; a slot holds func[n_lights], cdf[n_lights + 1], func_int
    uint32_t stride;         // 2 * n_lights + 2
    uint32_t capacity;
    uint32_t* new_list;      // voxel id of every slot, in claim order
    uint32_t* counters;      // [0] slots claimed, [1] slots computed, [2] pool-exhausted flag, [3] blocks finished (compute pass)
    const float* halton;     // 128 x {radical_inverse(0..4, i)}
};

enum : uint32_t { SP_CLAIMED = 0, SP_DONE = 1, SP_OVERFLOW = 2, SP_BLOCKS = 3 };

PH_DEV float sp_lerp(float t, float a, float b) { return (1.0f - t) * a + t * b; }  // pbrt/common.rs:167-173

// lookup's voxel coordinates (:170-181) with Bounds3::offset (bounds3.rs:153-168)
PH_DEV int32_t spatial_axis(float p, float lo, float hi, int32_t nv) {
    float o = p - lo;
    if (hi > lo) o = ph_div(o, hi - lo);
    const int32_t v = f2i_sat(o * (float)nv);
    return v < 0 ? 0 : (v > nv - 1 ? nv - 1 : v);
}
PH_DEV uint32_t spatial_voxel_of(const SpatialRec& sr, f3 p) {
    const int32_t x = spatial_axis(p.x, sr.lo[0], sr.hi[0], sr.nv[0]);
    const int32_t y = spatial_axis(p.y, sr.lo[1], sr.hi[1], sr.nv[1]);
    const int32_t z = spatial_axis(p.z, sr.lo[2], sr.hi[2], sr.nv[2]);
    return (uint32_t)((x * sr.nv[1] + y) * sr.nv[2] + z);
}

#define PH_SPATIAL_BLOCK 256
#define PH_SPATIAL_CHUNK 2048

// One block per new voxel (grid-stride over the slots claimed since the last pass).
__global__ __launch_bounds__(PH_SPATIAL_BLOCK) void spatial_compute_kernel(DeviceScene sc, SpatialRec sr) {
    __shared__ float buf[PH_SPATIAL_CHUNK];
    __shared__ float s_scalar[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = sc.n_lights;
    const uint32_t done = sr.counters[SP_DONE];
    uint32_t used = sr.counters[SP_CLAIMED];
    if (used > sr.capacity) used = sr.capacity;
    for (uint32_t slot = done + blockIdx.x; slot < used; slot += gridDim.x) {
        const uint32_t v = sr.new_list[slot];
        int32_t pi[3];
        pi[2] = (int32_t)(v % (uint32_t)sr.nv[2]);
        pi[1] = (int32_t)((v / (uint32_t)sr.nv[2]) % (uint32_t)sr.nv[1]);
        pi[0] = (int32_t)(v / ((uint32_t)sr.nv[2] * (uint32_t)sr.nv[1]));
        float lo[3], hi[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {  // voxel bounds (:95-106): Bounds3f::new(world.lerp(p0), world.lerp(p1))
            const float p0 = ph_div((float)pi[i], (float)sr.nv[i]), p1 = ph_div((float)(pi[i] + 1), (float)sr.nv[i]);
            const float a = sp_lerp(p0, sr.lo[i], sr.hi[i]), b = sp_lerp(p1, sr.lo[i], sr.hi[i]);
            lo[i] = pminf(a, b); hi[i] = pmaxf(a, b);
        }
        float* func = sr.pool + (size_t)slot * sr.stride;
        float* cdf = func + n;
        // ---- light_contrib[j] = sum_i y(Li) / pdf over the 128 sample points (:113-139), each light's sum in sample order
        for (uint32_t j = tid; j < n; j += PH_SPATIAL_BLOCK) {
            const LightRec& light = sc.lights[j];
            float acc = 0.0f;
#pragma unroll 1
            for (uint32_t i = 0; i < 128; i++) {
                const float* h = sr.halton + 5 * i;
                SurfHit intr;
                intr.p = mk3(sp_lerp(h[0], lo[0], hi[0]), sp_lerp(h[1], lo[1], hi[1]), sp_lerp(h[2], lo[2], hi[2]));
                intr.p_error = mk3(0, 0, 0); intr.wo = mk3(0, 0, 0); intr.n = mk3(0, 0, 0); intr.ns = mk3(0, 0, 0); intr.dpdu_s = mk3(0, 0, 0);
                intr.time = 0.0f; intr.prim = 0;
                const LiSample ls = light_sample_li(sc, light, intr, mk2(h[3], h[4]));
                if (ls.valid && ls.pdf > 0.0f) acc += ph_div(lum_y(ls.value), ls.pdf);
            }
            func[j] = acc;
        }
        __syncthreads();
        // ---- sum_contrib: `light_contrib.iter().sum()` is a left-to-right f32 sum (:144)
        float run = 0.0f;
        for (uint32_t base = 0; base < n; base += PH_SPATIAL_CHUNK) {
            const uint32_t cnt = n - base < PH_SPATIAL_CHUNK ? n - base : PH_SPATIAL_CHUNK;
            for (uint32_t k = tid; k < cnt; k += PH_SPATIAL_BLOCK) buf[k] = func[base + k];
            __syncthreads();
            if (tid == 0) for (uint32_t k = 0; k < cnt; k++) run += buf[k];
            __syncthreads();
        }
        if (tid == 0) {
            const float avg = ph_div(run, (float)(128ull * (unsigned long long)n));
            s_scalar[0] = avg > 0.0f ? 0.001f * avg : 1.0f;  // min_contrib (:146)
        }
        __syncthreads();
        const float min_contrib = s_scalar[0];
        // ---- clamp + Distribution1D::new (distribution_1d.rs:30-60): cdf[i] = cdf[i-1] + func[i-1] / n, sequentially
        const float fn = (float)n;
        run = 0.0f;
        if (tid == 0) cdf[0] = 0.0f;
        for (uint32_t base = 0; base < n; base += PH_SPATIAL_CHUNK) {
            const uint32_t cnt = n - base < PH_SPATIAL_CHUNK ? n - base : PH_SPATIAL_CHUNK;
            for (uint32_t k = tid; k < cnt; k += PH_SPATIAL_BLOCK) {
                const float c = pmaxf(func[base + k], min_contrib);  // max(contrib, min_contrib) (:149)
                func[base + k] = c; buf[k] = c;
            }
            __syncthreads();
            if (tid == 0) for (uint32_t k = 0; k < cnt; k++) { run = run + ph_div(buf[k], fn); buf[k] = run; }
            __syncthreads();
            for (uint32_t k = tid; k < cnt; k += PH_SPATIAL_BLOCK) cdf[base + k + 1] = buf[k];
            __syncthreads();
        }
        if (tid == 0) s_scalar[1] = run;
        __syncthreads();
        const float func_int = s_scalar[1];
        for (uint32_t k = 1 + tid; k <= n; k += PH_SPATIAL_BLOCK) cdf[k] = (func_int == 0.0f) ? ph_div((float)k, fn) : ph_div(cdf[k], func_int);
        if (tid == 0) {
            cdf[n + 1] = func_int;  // the slot's last float
            sr.vox_slot[v] = (int32_t)slot;
        }
        __syncthreads();
    }
    // the last block to finish publishes the new high-water mark (every block has read SP_DONE by then)
    if (tid == 0) {
        __threadfence();
        const uint32_t t = atomicAdd(&sr.counters[SP_BLOCKS], 1u);
        if (t == gridDim.x - 1) { sr.counters[SP_DONE] = used; sr.counters[SP_BLOCKS] = 0u; __threadfence(); }
    }
}

}  // namespace ph
