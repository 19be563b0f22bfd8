// Wavefront path tracer: SamplerIntegrator::render (core/src/integrator/sampler_integrator.rs:243-415) driving
// PathIntegrator::li (integrators/src/path.rs:103-284) as a sequence of data-parallel stages over a pool of paths.
//
//   K1 raygen   get_camera_sample + generate_ray_differential                      (sampler/mod.rs:45-53, perspective_camera.rs:144-204)
//   K2 extend   closest hit for extension rays AND the previous vertex's MIS rays     (traverse.h)
//   K5 shadow   any hit for the previous vertex's shadow rays                       (traverse.h)
//   K4 shade    resolve the previous vertex's direct lighting (K6), then li's loop body for the new vertex
//   K7 film     FilmTile::add_sample in the reference's accumulation order, Film::merge_film_tile in tile order
//
// No host synchronisation inside a chunk: every stage reads its item count from device memory (queue counters written by
// the previous stage with one wave-aggregated atomic per wave, __ballot/__popcll/__shfl), the traversal kernels are
// persistent, and a chunk is one fixed launch sequence.
//
// Float accumulation order is the reference's: L gets Le, then each vertex's Ld in bounce order (the MIS/shadow results of
// vertex k are folded in by the shade pass of iteration k+1 before anything of vertex k+1 is added); a pixel's FilmTile
// sums run over its samples in (pixel row-major, sample index) order.
#define PH_OUTLINE_MATH 1
#include "host_math.h"
#include "pt_device.h"
#include "scene_host.h"
#include "spatial.h"
#include "bsdf_general.h"
#include "texture.h"
#include "raysort.h"
#include "phase_clock.h"
#include "matsort.h"
#include <algorithm>
#include <cstdlib>

namespace ph {

enum : uint32_t { F_EXT = 1u, F_PSH = 2u, F_PMIS = 4u, F_SPEC = 8u, F_NODIFF = 16u };  // F_NODIFF: the ray was respawned at a "none" surface and has no differentials  // F_SPEC: the previous bounce sampled a specular lobe (path.rs:190)

struct IterCounters {  // one per wavefront iteration, zeroed at chunk start
    uint32_t n_cl, n_sh, n_live, head_cl, head_sh, pad[3];
};
struct DevStats { unsigned long long camera_rays, paths_total, paths_zero; };

struct TileInfo {  // one per local tile (tiles with index % parts == part, increasing index)
    int32_t tb[4];       // sample bounds of the tile (sampler_integrator.rs:331-336)
    int32_t pb[4];       // FilmTile pixel bounds (film/mod.rs:182-198)
    uint32_t px_off;     // first pixel of the tile in the rank's pixel list
    uint32_t tile_index; // global tile index
};

struct WfParams {
    // configuration
    CameraRec cam; SamplerRec sp;
    int32_t pixel_bounds[4];
    int32_t max_depth; float rr_threshold;
    uint32_t n_px;          // pixels in this rank's tiles
    uint32_t chunk_spp, s0; // samples [s0, s0+chunk_spp) of every pixel in this chunk
    uint32_t B;             // n_px * chunk_spp
    uint32_t identity_slots; // every pixel of the rank's tiles lies inside pixel_bounds
    // pixel list
    const int2* px_xy;
    // queues
    RayIn* rays_cl[2]; HitOut* hits_cl; RayIn* rays_sh; uint8_t* occ; uint32_t* live[2];
    IterCounters* ctr;
    DevStats* stats;
    // path state, in QUEUE order and double-buffered: round `it` reads buffer it & 1 at the path's position in that round's live list (streaming reads) and writes the
    // survivors' state to the other buffer at their position in the next round's list.  s_idx / s_L / s_beta travel with the compacted list; the data of a vertex's
    // pending light sample (s_A, s_f2, s_bold — known in the middle of the vertex code, before the survivor's slot is) is written at the WRITING thread's own position
    // and found again through s_prev.  (Rounds 1-2 addressed all of this by path id: scattered 16-byte records from six 2 GB arrays.)
    float4* s_L[2]; float4* s_beta[2]; float4* s_A[2]; float4* s_f2[2]; float4* s_bold[2]; uint4* s_idx[2]; uint32_t* s_prev[2];
    // per-sample records of the whole render: {L.rgb, p_film.x} {p_film.y}
    float4* rec_L; float* rec_py;
    uint8_t* px_rounded;    // per pixel of the rank's list: some sample's f32 film position rounded UP onto the next pixel's coordinate (film_tiles_kernel looks at those pixels' samples from the neighbour's side too)
    // light sampling: SpatialLightDistribution tables when enabled, else the scene-wide Distribution1D of DeviceScene
    SpatialRec spatial;
    // materials that evaluate a texture per hit (texture.h): device copy of `cam` for the out-of-line evaluation
    const CameraRec* cam_dev; uint32_t textured;
    TexOut* tex_out;     // texture pass -> shade pass, one record per path of the chunk
    uint32_t any_rt;     // some TranslucentMaterial decides per hit whether it has a BSDF at all (MaterialRec::rt_mode): the texture pass runs first and says so in TexOut::bumped
    // ray binning between rounds (raysort.h): the bin key of every ray the shade pass emits, next to the ray
    RaySortGrid sort_grid; uint32_t* keys_cl; uint32_t* keys_sh;
    // shade-side work queues (matsort.h): this round's list positions grouped by material key, and the bins' starts; null = the texture / light-distribution / shade passes walk the list in queue order
    const uint32_t* m_order; const uint32_t* m_bins;
    unsigned long long* phase;   // PH_PHASE_CLOCK builds: the shade kernel's phase tallies (phase_clock.h)
};

PH_DEV uint32_t wave_alloc(uint32_t* ctr, bool want) {
    const uint64_t m = __ballot(want);
    if (m == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    return base + (uint32_t)__popcll(m & lt);
}

PH_DEV SamplerCursor cursor_for(const DeviceScene& sc, const SamplerRec& sp, int px, int py, uint32_t s, uint32_t dim, const HaltonLds* lds = nullptr) {
    SamplerCursor c; c.px = px; c.py = py; c.dim = dim; c.lds = lds;
    if (sp.kind == 0) c.index = (uint64_t)halton_pixel_offset(sp, px, py) + (uint64_t)s * sp.sample_stride;  // halton.rs:143
    else c.index = sobol_interval_to_index(sc, (uint32_t)sp.log2_resolution, s, px - sp.bounds[0], py - sp.bounds[1]);
    return c;
}
PH_DEV void store_ray(RayIn* dst, const RayIn& r) {
    float4* p = reinterpret_cast<float4*>(dst);
    p[0] = make_float4(r.ox, r.oy, r.oz, r.t_max); p[1] = make_float4(r.dx, r.dy, r.dz, r.time);
}
PH_DEV RayIn load_ray(const RayIn* src) {
    const float4* p = reinterpret_cast<const float4*>(src);
    float4 a = p[0], b = p[1];
    RayIn r; r.ox = a.x; r.oy = a.y; r.oz = a.z; r.t_max = a.w; r.dx = b.x; r.dy = b.y; r.dz = b.z; r.time = b.w; return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// K1: one thread per (pixel, sample) of the chunk
__global__ __launch_bounds__(256) void raygen_kernel(DeviceScene sc, WfParams w) {
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = pid < w.B;
    RayIn ray;
    f2 lens_keep = mk2(0.0f, 0.0f);
    if (active) {
        const uint32_t pix = pid / w.chunk_spp, j = pid - pix * w.chunk_spp, s = w.s0 + j;  // a wave = consecutive samples of one pixel (coherent rays)
        const int2 xy = w.px_xy[pix];
        // pixels outside the integrator's pixel_bounds are skipped after start_pixel (sampler_integrator.rs:348-350)
        active = xy.x >= w.pixel_bounds[0] && xy.x < w.pixel_bounds[2] && xy.y >= w.pixel_bounds[1] && xy.y < w.pixel_bounds[3];
        const size_t gsi = (size_t)s * w.n_px + pix;  // sample records are [sample][pixel]: film_tiles_kernel reads them coalesced
        if (active) {
            SamplerCursor c = cursor_for(sc, w.sp, xy.x, xy.y, s, 0);
            f2 fs = get_2d(sc, w.sp, c);
            f2 p_film = mk2((float)xy.x + fs.x, (float)xy.y + fs.y);
            float time = get_1d(sc, w.sp, c);
            f2 lens = get_2d(sc, w.sp, c);
            generate_camera_ray(w.cam, p_film, time, lens, ray);
            w.rec_L[gsi] = make_float4(0.0f, 0.0f, 0.0f, p_film.x);
            w.rec_py[gsi] = p_film.y;
            if (p_film.x == (float)(xy.x + 1) || p_film.y == (float)(xy.y + 1)) w.px_rounded[pix] = 1u;   // (every writer writes the same byte)
            lens_keep = lens;
        } else {
            w.rec_L[gsi] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0x7fc00000u));  // NaN p_film.x marks "no sample"
            w.rec_py[gsi] = 0.0f;
        }
    }
    // common case (pixel_bounds covers the tiles): slot = pid, counters preset by the host -> no atomics at all
    const uint32_t slot = w.identity_slots ? pid : wave_alloc(&w.ctr[0].n_cl, active);
    const uint32_t lslot = w.identity_slots ? pid : wave_alloc(&w.ctr[0].n_live, active);
    if (active) {
        store_ray(w.rays_cl[0] + slot, ray);
        w.live[0][lslot] = pid;
        w.s_idx[0][lslot] = make_uint4(slot, 0u, 0u, F_EXT | (0u << 8) | (5u << 16));  // bounces 0, next sampler dimension 5
        w.s_L[0][lslot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        w.s_beta[0][lslot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // beta, eta_scale
        w.s_prev[0][lslot] = lslot;
        if (w.textured && w.cam.lens_radius > 0.0f) w.s_A[0][lslot] = make_float4(lens_keep.x, lens_keep.y, 0.0f, 0.0f);  // for the first hit's ray differentials (s_A is idle until then)
        if (!w.identity_slots) atomicAdd(&w.stats->camera_rays, 1ull);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// SpatialLightDistribution::lookup, first half (spatial.rs:166-236): every path that is about to sample a light at a new
// vertex names its voxel; the first path to touch a voxel claims a pool slot for it (spatial_compute_kernel fills it).
__global__ __launch_bounds__(256) void spatial_mark_kernel(DeviceScene sc, WfParams w, int it) {
    const uint32_t n_live = w.ctr[it].n_live;
    const SpatialRec& sr = w.spatial;
    // (with the work queues: only the paths in front of the "emission only" and "no new vertex" bins have a vertex that samples a light)
    const uint32_t n_work = w.m_order ? w.m_bins[sc.ms_key_emit] : n_live;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_work; j += gridDim.x * blockDim.x) {
        const uint32_t i = w.m_order ? w.m_order[j] : j;   // the path's position in the round's list; the texture pass's record of it is at j
        const uint4 idx4 = w.s_idx[it & 1][i];
        const uint32_t flags = idx4.w & 0xffu, bounces = (idx4.w >> 8) & 0xffu;
        if (!(flags & F_EXT) || (int)bounces >= w.max_depth) continue;
        const float4* hp = reinterpret_cast<const float4*>(w.hits_cl + idx4.x);
        const float4 h0 = hp[0];
        if (__float_as_uint(h0.y) == 0xFFFFFFFFu) continue;
        const float4 h1 = hp[1];
        if ((__float_as_uint(h1.w) & 7u) == 7u) continue;  // Material "none": the surface is skipped before the lookup (path.rs:146-150)
        const float4* tp = reinterpret_cast<const float4*>(sc.tris + __float_as_uint(h1.y));
        const float4 a = tp[0], b = tp[1], c = tp[2];
        // ... and so is a hit for which this round's texture pass found no BSDF (a TranslucentMaterial whose reflect and transmit are black there)
        if (w.any_rt && sc.materials[sc.meshes[__float_as_uint(c.w)].material].rt_mode && (w.tex_out[j].bumped & PH_TEXOUT_NULL_BSDF)) continue;
        f3 p = h0.z * mk3(a.x, a.y, a.z) + h0.w * mk3(b.x, b.y, b.z) + h1.x * mk3(c.x, c.y, c.z);  // = SurfHit::p (make_surface_hit_tv)
        const uint32_t inst = __float_as_uint(h1.z);
        if (inst != 0u && !(sc.instances[inst - 1u].flags & PH_INST_IDENTITY)) {  // world-space p of an instanced hit (make_surface_hit_any)
            f3 pe; p = xf_point_abs_err(sc.instances[inst - 1u].i2w, p, mk3(0.0f, 0.0f, 0.0f), pe);
        }
        const uint32_t v = spatial_voxel_of(sr, p);
        if (sr.vox_slot[v] == -1 && atomicCAS(&sr.vox_slot[v], -1, -2) == -1) {
            const uint32_t slot = atomicAdd(&sr.counters[SP_CLAIMED], 1u);
            if (slot < sr.capacity) sr.new_list[slot] = v;
            else sr.counters[SP_OVERFLOW] = 1u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Texture pass: one thread per live path whose new vertex lies on a material with per-hit textures or a bump map.  It rebuilds the surface
// interaction, the texture context (uv, dp/du, dp/dv, camera-ray differentials), runs Material::bump and evaluates the textured lobe colours,
// and leaves the results in tex_out for the shade pass.  A pass of its own so that the evaluator's registers and calls stay out of the shade kernel.
// With the work queues (matsort.h, round 4) its work is the PREFIX of the sorted list — order[0 .. bins[ms_tex_keys]) — in material order: full waves, one material's
// texture programs per wave (round 3 walked the whole live list and `continue`d past three paths in four on configs[4]); record j of tex_out belongs to order[j].
// SIMPLE: the scene's textures are constants, image maps, scale and mix only: the evaluator is compiled without the procedural classes and fits more waves.
// CAMERA: round 0, whose rays are the camera rays — the only ones that carry differentials (path.rs:107, sampler_integrator.rs:358); every later round's vertices are
//   evaluated without (NODIFF evaluator, texture.h: no EWA / trilinear code, no differentials of the camera ray) — a leaner kernel for five rounds of six.
template <bool SIMPLE, bool CAMERA, int WAVES = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void texture_kernel(DeviceScene sc, WfParams w, int it) {
    const uint32_t n_work = w.m_order ? w.m_bins[sc.ms_tex_keys] : w.ctr[it].n_live;
    const uint32_t* live_in = w.live[it & 1];
    const RayIn* rays_in = w.rays_cl[it & 1];
    if (!SIMPLE) noise_lds_fill();   // the procedural classes' Perlin table, into LDS (texture.h)
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __shared__ unsigned long long phc_lds[3 * PHC_N];
    if (threadIdx.x < 3 * PHC_N) phc_lds[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    PHC_BEGIN(3);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_work; j += gridDim.x * blockDim.x) {
        const uint32_t i = w.m_order ? w.m_order[j] : j;   // the path's position in the round's list
        const uint4 idx4 = w.s_idx[it & 1][i];
        const uint32_t flags = idx4.w & 0xffu, bounces = (idx4.w >> 8) & 0xffu;
        if (!(flags & F_EXT) || (int)bounces >= w.max_depth) continue;
        const float4* hp = reinterpret_cast<const float4*>(w.hits_cl + idx4.x);
        const float4 h0 = hp[0];
        if (__float_as_uint(h0.y) == 0xFFFFFFFFu) continue;
        const float4 h1 = hp[1];
        PHC_BEGIN(0);
        const RayIn ray = load_ray(rays_in + idx4.x);
        const f3 rd = mk3(ray.dx, ray.dy, ray.dz);
        MeshRec m;
        const SurfHit si = make_surface_hit_any(sc, rd, ray.time, __float_as_uint(h1.y), __float_as_uint(h1.z), h0.z, h0.w, h1.x, m);
        const MaterialRec& mr = sc.materials[m.material];
        if (mr.none || !(mr.textured || mr.bump_tex1)) continue;
        const uint32_t camera_ray = (CAMERA && bounces == 0u && !(flags & F_NODIFF)) ? 1u : 0u;  // only camera rays carry differentials
        f2 p_film = mk2(0.0f, 0.0f), lens = mk2(0.0f, 0.0f);
        if (CAMERA && camera_ray) {
            const uint32_t pid = live_in[i];
            const uint32_t ppix = pid / w.chunk_spp;
            const size_t gsi = (size_t)(w.s0 + (pid - ppix * w.chunk_spp)) * w.n_px + ppix;
            p_film = mk2(w.rec_L[gsi].w, w.rec_py[gsi]);
            if (w.cam.lens_radius > 0.0f) { const float4 la = w.s_A[it & 1][w.s_prev[it & 1][i]]; lens = mk2(la.x, la.y); }
        }
        const TexCtx ctx = hit_tex_ctx(sc.self, w.cam_dev, w.sp.spp, __float_as_uint(h1.y), __float_as_uint(h1.z), mk3(h0.z, h0.w, h1.x), si.p, si.n,
                                       mk3(ray.ox, ray.oy, ray.oz), rd, p_film, lens, camera_ray);
        PHC_END(0);
        TexOut out;
        out.bumped = 0u; out.lambert = 0u;
        out.ns[0] = si.ns.x; out.ns[1] = si.ns.y; out.ns[2] = si.ns.z; out.dpdu_s[0] = si.dpdu_s.x; out.dpdu_s[1] = si.dpdu_s.y; out.dpdu_s[2] = si.dpdu_s.z;
        if (mr.bump_tex1) {
            PHC_BEGIN(1);
            BumpIn bi; bi.tex = mr.bump_tex1 - 1u; bi.tri_index = __float_as_uint(h1.y); bi.inst = __float_as_uint(h1.z); bi.bary = mk3(h0.z, h0.w, h1.x);
            bi.p = si.p; bi.n = si.n; bi.ns = si.ns; bi.dpdu_s = si.dpdu_s; bi.c = ctx;
            BumpOut bo;
            hit_bump<SIMPLE, !CAMERA>(sc.self, &bi, &bo);
            out.ns[0] = bo.ns.x; out.ns[1] = bo.ns.y; out.ns[2] = bo.ns.z; out.dpdu_s[0] = bo.dpdu_s.x; out.dpdu_s[1] = bo.dpdu_s.y; out.dpdu_s[2] = bo.dpdu_s.z;
            out.bumped = 1u;
            PHC_END(1);
        }
        for (int k = 0; k < PH_HIT_COLS; k++) out.col[k][0] = out.col[k][1] = out.col[k][2] = out.col[k][3] = 0.0f;
        if (mr.textured) { PHC_BEGIN(2); eval_lobe_colours<SIMPLE, !CAMERA>(sc.self, mr, sc.lobes + mr.lobe_base, mr.n_lobes, ctx, out); PHC_END(2); }

        // only what the shade pass reads for this material: the two header quads if it needs them, then the colour slots in use (a matte with an image map: 16 of the 128 bytes)
        float4* dst = reinterpret_cast<float4*>(w.tex_out + j);   // by position in the round's (sorted) walk: the shade pass's thread j reads it from there
        const float4* src = reinterpret_cast<const float4*>(&out);
        if (mr.tex_hdr) { dst[0] = src[0]; dst[1] = src[1]; }
        for (uint32_t k = 0; k < mr.tex_cols; k++) dst[2 + k] = src[2 + k];
    }
    PHC_END(3);
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
    if (threadIdx.x < 3 * PHC_N && w.phase) atomicAdd(w.phase + 3 * PHC_N + threadIdx.x, phc_lds[threadIdx.x]);   // (behind the shade kernel's tallies)
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// K4 (+K6): one thread per live path.
// Queue appends are aggregated per BLOCK: rays are staged in LDS the moment they are known (which keeps them out of the
// register file during the long vertex computation), then one thread per queue claims the block's slots with a single
// atomic and every thread copies its rays to consecutive slots.  Per-wave atomics on the three queue counters were the
// limiter of the first version (≈3 M returning atomics per frame on three addresses; one address sustains ≈88 per µs).
// BSDF front ends of the two shade_kernel instantiations: GEN = false keeps the one-lobe matte code (pt_device.h) for scenes made
// of MatteMaterial only — the headline workload —, GEN = true walks the general lobe list (bsdf_general.h).
template <bool GEN> struct BsdfOps;
template <> struct BsdfOps<false> {
    using T = Bsdf;
    static PH_DEV T make(const DeviceScene& sc, const SurfHit& si, uint32_t mat) { return make_bsdf(sc, si, mat); }
    static PH_DEV bool has_non_specular(const T& b) { return b.has_bxdf; }
    static PH_DEV spec f_ns(const T& b, f3 wo, f3 wi) { return bsdf_f(b, wo, wi); }
    static PH_DEV float pdf_ns(const T& b, f3 wo, f3 wi) { return bsdf_pdf(b, wo, wi); }
    static PH_DEV void sample_ns(const T& b, f3 wo, f2 u, spec& f, float& pdf, f3& wi) { bsdf_sample_f(b, wo, u, f, pdf, wi); }
    static PH_DEV void sample_all(const T& b, f3 wo, f2 u, spec& f, float& pdf, f3& wi, uint32_t& type) { bsdf_sample_f(b, wo, u, f, pdf, wi); type = BX_REFL | BX_DIFF; }
    static PH_DEV float eta(const T&) { return 1.0f; }
    static PH_DEV void apply_textures(T& b, const TexOut* to, const MaterialRec& mr) {  // MatteMaterial: Kd and / or sigma of this hit
        if (mr.kd_tex1) { const spec kd = mks(to->col[0][0], to->col[0][1], to->col[0][2]); b.r = kd; b.has_bxdf = !is_black(kd); }
        if (mr.sigma_tex1) { b.oren = (to->lambert & 1u) == 0u; b.a = to->col[0][3]; b.b = to->col[1][3]; }
    }
};
template <> struct BsdfOps<true> {
    using T = GBsdf;
    static PH_DEV T make(const DeviceScene& sc, const SurfHit& si, uint32_t mat) { return make_gbsdf(sc, si, mat); }
    static PH_DEV bool has_non_specular(const T& b) { return bsdf_num_components(b, BX_ALL & ~BX_SPEC) > 0u; }
    static PH_DEV spec f_ns(const T& b, f3 wo, f3 wi) { return bsdf_f(b, wo, wi, BX_ALL & ~BX_SPEC); }   // estimate_direct: specular = false (common.rs:157-161)
    static PH_DEV float pdf_ns(const T& b, f3 wo, f3 wi) { return bsdf_pdf(b, wo, wi, BX_ALL & ~BX_SPEC); }
    static PH_DEV void sample_ns(const T& b, f3 wo, f2 u, spec& f, float& pdf, f3& wi) { uint32_t t; bsdf_sample_f(b, wo, u, BX_ALL & ~BX_SPEC, f, pdf, wi, t); }
    static PH_DEV void sample_all(const T& b, f3 wo, f2 u, spec& f, float& pdf, f3& wi, uint32_t& type) { bsdf_sample_f(b, wo, u, BX_ALL, f, pdf, wi, type); }
    static PH_DEV float eta(const T& b) { return b.eta; }
    // the hit's own lobe list: which template lobes it holds; their textured fields are read from the texture pass's record when a lobe is used (bsdf_general.h)
    static PH_DEV void apply_textures(T& b, const TexOut* to, const MaterialRec& mr) {
        b.keep = hit_lobe_mask(mr, b.lobes, b.n, to, b.eta);
        b.hit = to;
    }
};

#define PH_TEX_SIMPLE_WAVES 3   // waves per SIMD the image-map-only texture pass is compiled for: textured configs[1] 54.3 ms per frame at 3, 56.4 at 4 (254 spilled registers), 55.7 at 2 (gpurun r02q)
#define PH_SHADE_BLOCK 256
// Waves per SIMD the shade kernels are compiled for.  40 KB of LDS per block allow 4 blocks per CU; the one-lobe kernel fits 128 VGPRs with
// 51 spilled registers and gains 6 % from the fourth wave, the general-BSDF kernel would spill 181 and loses, so it stays at 3; so do the
// texture variants, whose out-of-line calls keep many values live (textured matte: 279 spilled registers at 4 waves, 45 at 3; 19.5 -> 15.9 ms).
#ifndef PH_SHADE_GEN_TEX_WAVES
#define PH_SHADE_GEN_TEX_WAVES 3
#endif
#define PH_SHADE_WAVES(GEN, TEX) ((GEN) && (TEX) ? PH_SHADE_GEN_TEX_WAVES : (((GEN) || (TEX)) ? 3 : 4))
#define PH_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(PH_SHADE_WAVES(GEN, TEX), PH_SHADE_WAVES(GEN, TEX))))
template <bool GEN, bool TEX = false>
__global__ __launch_bounds__(PH_SHADE_BLOCK) PH_SHADE_ATTR void shade_kernel(DeviceScene sc, WfParams w, int it) {
    using BO = BsdfOps<GEN>;
    __shared__ float4 stage[3][2][PH_SHADE_BLOCK];           // [ext, mis, shadow][ray halves][thread]
    __shared__ uint32_t wave_cnt[3][PH_SHADE_BLOCK / 64];   // [cl, sh, live][wave]
    __shared__ uint32_t q_base[3];
    __shared__ HaltonLds halton_lds;
    __shared__ float s_u[8][PH_SHADE_BLOCK];  // the (up to) 8 sampler dimensions a vertex can consume, drawn by ONE loop
    const uint32_t n_live = w.ctr[it].n_live;
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __shared__ unsigned long long phc_lds[3 * PHC_N];
    if (threadIdx.x < 3 * PHC_N) phc_lds[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    PHC_BEGIN(10);
    const HaltonLds* hl = nullptr;
    if (w.sp.kind == 0 && blockIdx.x * blockDim.x < n_live) { halton_lds_fill(&halton_lds, sc); hl = &halton_lds; }
    __syncthreads();
    const RayIn* rays_in = w.rays_cl[it & 1];
    RayIn* rays_out = w.rays_cl[(it + 1) & 1];
    const uint32_t* live_in = w.live[it & 1];
    uint32_t* live_out = w.live[(it + 1) & 1];
    IterCounters* next = w.ctr + it + 1;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // "Integrator/Zero-radiance paths" counters (path.rs:18-24): kept in registers and flushed once per block — a global
    // atomic per vertex on one address costs more than the whole vertex (one address sustains ~88 atomics/us).
    uint32_t n_paths_total = 0, n_paths_zero = 0;
    __shared__ uint32_t blk_stats[2];
    if (tid < 2) blk_stats[tid] = 0;

    for (uint32_t base = blockIdx.x * blockDim.x; base < n_live; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + tid;
        const bool active = i < n_live;
        uint32_t pid = 0, flags = 0, bounces = 0, dim = 0;
        spec L = mks1(0.0f), beta = mks1(1.0f);
        float pick_pdf = 0.0f, eta_scale = 1.0f;
        bool want_ext = false, want_mis = false, want_sh = false;

        if (active) {
            // this thread's path: its position in the round's list.  With the work queues (matsort.h) the walk goes through `order`: a wave holds paths of ONE material (its
            // lobe list, texture slots and light-sampling branches are then the wave's), paths with nothing to shade sit at the end.  Which thread handles which path is free:
            // paths are independent and the film is accumulated by sample index.  (Rounds 1 - 3 sorted the 256 paths of a block by material class in LDS: ~9 paths per class.)
            PHC_BEGIN(0);
            const uint32_t pos = w.m_order ? w.m_order[i] : i;
            pid = live_in[pos];
            const uint4 idx4 = w.s_idx[it & 1][pos];
            flags = idx4.w & 0xffu; bounces = (idx4.w >> 8) & 0xffu; dim = idx4.w >> 16;
            const float4 L4 = w.s_L[it & 1][pos], b4 = w.s_beta[it & 1][pos];
            L = mks(L4.x, L4.y, L4.z); beta = mks(b4.x, b4.y, b4.z);
            if (GEN) eta_scale = b4.w;

            // ---- K6: finish uniform_sample_one_light of the previous vertex (integrator/common.rs:196-221, 276-295, 132) --------
            if (flags & (F_PSH | F_PMIS)) {
                const uint32_t prev = w.s_prev[it & 1][pos];   // where the thread that made this vertex's light sample left it
                const float4 A4 = w.s_A[it & 1][prev], O4 = w.s_bold[it & 1][prev];
                spec est = mks1(0.0f);
                if (flags & F_PSH) {
                    if (!w.occ[idx4.z]) est = est + mks(A4.x, A4.y, A4.z);
                }
                if (flags & F_PMIS) {
                    const float4 F4 = w.s_f2[it & 1][prev];
                    const uint32_t light_num = __float_as_uint(F4.w);
                    const RayIn mr = load_ray(rays_in + idx4.y);
                    const float4* hp = reinterpret_cast<const float4*>(w.hits_cl + idx4.y);
                    const float4 h0 = hp[0];
                    const uint32_t hprim = __float_as_uint(h0.y);
                    const f3 wi = mk3(mr.dx, mr.dy, mr.dz);
                    spec li2 = mks1(0.0f);
                    if (hprim != 0xFFFFFFFFu) {
                        const LightRec& lt = sc.lights[light_num];
                        if (lt.type == PH_L_AREA && lt.prim == hprim) {  // the hit primitive's area light IS the sampled light
                            const float4 h1 = hp[1];
                            MeshRec m;
                            SurfHit lh = make_surface_hit_any(sc, wi, mr.time, __float_as_uint(h1.y), __float_as_uint(h1.z), h0.z, h0.w, h1.x, m);
                            li2 = area_L(lt, lh.n, -wi);  // SurfaceInteraction::le (surface_interaction.rs:283-289)
                        }
                    } else li2 = light_le<TEX>(sc, sc.lights[light_num], wi);
                    if (!is_black(li2)) est = est + mks(F4.x, F4.y, F4.z) * li2 * mks1(1.0f) * A4.w / O4.w;  // f*li*tr*weight/scattering_pdf
                }
                const spec ldv = mks(O4.x, O4.y, O4.z) * (est / L4.w);  // beta * (estimate / light_pdf)  (path.rs:165)
                if (is_black(ldv)) n_paths_zero++;
                L = L + ldv;
                flags &= ~(F_PSH | F_PMIS);
            }
            PHC_END(0);

            // ---- the new vertex: body of li's loop (path.rs:116-279) ---------------------------------------------------------------
            if (flags & F_EXT) {
                flags &= ~F_EXT;
                const RayIn ray = load_ray(rays_in + idx4.x);
                const float4* hp = reinterpret_cast<const float4*>(w.hits_cl + idx4.x);
                const float4 h0 = hp[0];
                const uint32_t hprim = __float_as_uint(h0.y);
                const bool found = hprim != 0xFFFFFFFFu;
                const f3 rd = mk3(ray.dx, ray.dy, ray.dz);
                const bool emit = bounces == 0 || (GEN && (flags & F_SPEC));  // `bounces == 0 || specular_bounce` (path.rs:116)
                if (!found) {
                    if (emit)
                        for (uint32_t k = 0; k < sc.n_infinite; k++) L = L + beta * light_le<TEX>(sc, sc.lights[sc.infinite_lights[k]], rd);
                } else {
                    const float4 h1 = hp[1];
                    MeshRec m;
                    PHC_BEGIN(1);
                    SurfHit si = make_surface_hit_any(sc, rd, ray.time, __float_as_uint(h1.y), __float_as_uint(h1.z), h0.z, h0.w, h1.x, m);
                    if (emit) {
                        if (m.first_light >= 0) L = L + beta * area_L(sc.lights[(uint32_t)m.first_light + (hprim - m.tri_base)], si.n, -rd);
                        else L = L + beta * mks1(0.0f);
                    }
                    PHC_END(1);
                    bool no_bsdf = sc.materials[m.material].none != 0u;
                    if (TEX && !no_bsdf && sc.materials[m.material].rt_mode) no_bsdf = (w.tex_out[i].bumped & PH_TEXOUT_NULL_BSDF) != 0u;   // translucent.rs:72-74, decided by this hit's textures
                    if ((int)bounces < w.max_depth && no_bsdf) {
                        // null BSDF (Material "none"): `*ray = isect.spawn_ray(&ray.d); continue;` — bounces, the sampler dimension and the
                        // specular flag stay as they are (path.rs:142-150)
                        const RayIn re = spawn_ray(si, rd);
                        flags |= F_EXT | F_NODIFF; want_ext = true;
                        stage[0][0][tid] = make_float4(re.ox, re.oy, re.oz, re.t_max);
                        stage[0][1][tid] = make_float4(re.dx, re.dy, re.dz, re.time);
                    } else if ((int)bounces < w.max_depth) {
                        const uint32_t ppix = pid / w.chunk_spp;
                        const int2 xy = w.px_xy[ppix];
                        bool tex_hit = false;
                        if (TEX) {  // the texture pass (texture_kernel) left this vertex's bumped frame and textured colours in tex_out[i] (i = this thread's place in the round's walk)
                            const MaterialRec& mr = sc.materials[m.material];
                            tex_hit = mr.textured != 0u || mr.bump_tex1 != 0u;
                            if (tex_hit && mr.bump_tex1) {  // Material::bump: the BSDF is made on the bumped frame
                                const float4* tp = reinterpret_cast<const float4*>(w.tex_out + i);
                                const float4 f0 = tp[0], f1 = tp[1];
                                si.ns = mk3(f0.x, f0.y, f0.z); si.dpdu_s = mk3(f1.x, f1.y, f1.z);
                            }
                        }
                        PHC_BEGIN(2);
                        typename BO::T bsdf = BO::make(sc, si, m.material);
                        if (TEX && tex_hit) {
                            const MaterialRec& mr = sc.materials[m.material];
                            if (mr.textured)
                                BO::apply_textures(bsdf, w.tex_out + i, mr);
                        }
                        PHC_END(2);
                        PHC_BEGIN(3);
                        SamplerCursor cur = cursor_for(sc, w.sp, xy.x, xy.y, w.s0 + (pid - ppix * w.chunk_spp), dim, hl);
                        // Draw the next 8 dimensions in one (not unrolled) loop: light pick 1D, u_light 2D, u_scattering 2D, BSDF 2D,
                        // Russian roulette 1D.  li consumes a prefix of them that depends on the vertex (A4 ledger); which VALUE lands in
                        // which role is decided below exactly as get_1d/get_2d would, only the evaluation is hoisted (one copy of the
                        // radical-inverse code instead of ten keeps the kernel inside the instruction cache).
#pragma unroll 1
                        for (uint32_t k = 0; k < 8; k++) s_u[k][tid] = sampler_dim(sc, w.sp, cur, dim + k);
                        PHC_END(3);
                        uint32_t c = 0;  // dimensions consumed so far at this vertex
                        if (BO::has_non_specular(bsdf)) {  // num_components(all & !SPECULAR) > 0 (path.rs:161-172)
                            n_paths_total++;
                            // uniform_sample_one_light (integrator/common.rs:89-133)
                            if (sc.n_lights > 0) {
                                PHC_BEGIN(4);
                                const float sample = s_u[0][tid]; c = 1;
                                // sample_discrete; with a single light the CDF is {0, 1} and the answer is 0 for every sample in [0,1):
                                // keeping that case wave-uniform lets the light record be fetched with scalar loads
                                const float* ld_func = sc.ld_func; const float* ld_cdf = sc.ld_cdf; float ld_func_int = sc.ld_func_int;
                                if (w.spatial.enabled) {  // light_distribution.lookup(&isect.hit.p) (path.rs:156-157)
                                    const int32_t slot = w.spatial.vox_slot[spatial_voxel_of(w.spatial, si.p)];
                                    if (slot >= 0) {
                                        ld_func = w.spatial.pool + (size_t)slot * w.spatial.stride;
                                        ld_cdf = ld_func + sc.n_lights; ld_func_int = ld_cdf[sc.n_lights + 1];
                                    } else ld_func_int = 0.0f;  // pool exhausted: the render call fails (ERR_OOM), nothing is returned
                                }
                                const uint32_t light_num = (sc.n_lights == 1) ? 0u : find_interval_cdf(ld_cdf, sc.n_lights + 1, sample);
                                pick_pdf = ld_func_int > 0.0f ? ph_div(ld_func[light_num], ld_func_int * (float)sc.n_lights) : 0.0f;
                                PHC_END(4);
                                if (pick_pdf != 0.0f) {
                                    const LightRec& light = sc.lights[light_num];
                                    const f2 u_light = mk2(s_u[1][tid], s_u[2][tid]), u_scatter = mk2(s_u[3][tid], s_u[4][tid]); c = 5;
                                    const bool is_delta = light.type == PH_L_DISTANT || light.type == PH_L_POINT || light.type == PH_L_SPOT || light.type == PH_L_PROJECTION || light.type == PH_L_GONIO;
                                    float w2 = 0.0f, spdf_store = 0.0f;
                                    spec A = mks1(0.0f);
                                    // estimate_direct (integrator/common.rs:146-299), specular = false, handle_media = false
                                    {
                                        PHC_BEGIN(5);
                                        const LiSample ls = light_sample_li<TEX>(sc, light, si, u_light);
                                        PHC_END(5);
                                        PHC_BEGIN(6);
                                        if (ls.valid && ls.pdf > 0.0f && !is_black(ls.value)) {
                                            const spec f = BO::f_ns(bsdf, si.wo, ls.wi) * abs_dot(ls.wi, si.ns);
                                            const float scattering_pdf = BO::pdf_ns(bsdf, si.wo, ls.wi);
                                            if (!is_black(f)) {
                                                const RayIn rs = spawn_ray_to_hit(si, ls.vp, ls.vperr, ls.vn);  // VisibilityTester::unoccluded
                                                stage[2][0][tid] = make_float4(rs.ox, rs.oy, rs.oz, rs.t_max);
                                                stage[2][1][tid] = make_float4(rs.dx, rs.dy, rs.dz, rs.time);
                                                want_sh = true;
                                                if (is_delta) A = f * ls.value / ls.pdf;
                                                else A = f * ls.value * power_heuristic1(ls.pdf, scattering_pdf) / ls.pdf;
                                                flags |= F_PSH;
                                            }
                                        }
                                        PHC_END(6);
                                    }
                                    if (!is_delta) {
                                        PHC_BEGIN(7);
                                        spec f1; float spdf; f3 wi2;
                                        BO::sample_ns(bsdf, si.wo, u_scatter, f1, spdf, wi2);  // never specular with these flags: sampled_specular = false
                                        const spec f = f1 * abs_dot(wi2, si.ns);
                                        if (!is_black(f) && spdf > 0.0f) {
                                            const float lp = light_pdf_li<TEX>(sc, light, si, wi2);
                                            if (lp != 0.0f) {  // lp == 0 -> `return ld` with the light-sampling part only
                                                w2 = power_heuristic1(spdf, lp);
                                                spdf_store = spdf;
                                                const RayIn rm = spawn_ray(si, wi2);
                                                stage[1][0][tid] = make_float4(rm.ox, rm.oy, rm.oz, rm.t_max);
                                                stage[1][1][tid] = make_float4(rm.dx, rm.dy, rm.dz, rm.time);
                                                want_mis = true;
                                                w.s_f2[(it + 1) & 1][i] = make_float4(f.r, f.g, f.b, __uint_as_float(light_num));
                                                flags |= F_PMIS;
                                            }
                                        }
                                        PHC_END(7);
                                    }
                                    if (flags & (F_PSH | F_PMIS)) {
                                        w.s_A[(it + 1) & 1][i] = make_float4(A.r, A.g, A.b, w2);   // at THIS thread's position (coalesced); the survivor's s_prev points here
                                        w.s_bold[(it + 1) & 1][i] = make_float4(beta.r, beta.g, beta.b, spdf_store);
                                    }
                                }
                            }
                            if (!(flags & (F_PSH | F_PMIS))) n_paths_zero++;  // ld is black
                        }
                        // sample the BSDF for the next direction (path.rs:174-206)
                        PHC_BEGIN(8);
                        const f2 u = mk2(s_u[c][tid], s_u[c + 1][tid]); c += 2;
                        spec f; float pdf; f3 wi; uint32_t stype;
                        BO::sample_all(bsdf, -rd, u, f, pdf, wi, stype);
                        flags &= ~F_SPEC;
                        if (!(is_black(f) || pdf == 0.0f)) {
                            beta = beta * (f * abs_dot(wi, si.ns) / pdf);
                            if (GEN) {
                                if (stype & BX_SPEC) flags |= F_SPEC;  // specular_bounce (path.rs:190)
                                if ((stype & BX_SPEC) && (stype & BX_TRANS)) {  // radiance scaling across a refraction (path.rs:192-203)
                                    const float eta = BO::eta(bsdf);
                                    eta_scale *= dot(-rd, si.n) > 0.0f ? eta * eta : ph_div(1.0f, eta * eta);
                                }
                            }
                            const RayIn re = spawn_ray(si, wi);
                            bool cont = true;
                            const spec rr_beta = beta * eta_scale;  // eta_scale stays 1 without specular transmission
                            if (max_component_value(rr_beta) < w.rr_threshold && bounces > 3) {  // path.rs:264-276
                                const float q = pmaxf(0.05f, 1.0f - max_component_value(rr_beta));
                                const float rr = s_u[c][tid]; c += 1;
                                if (rr < q) cont = false;
                                else beta = beta / (1.0f - q);
                            }
                            if (cont) {
                                bounces += 1; flags |= F_EXT; want_ext = true;
                                stage[0][0][tid] = make_float4(re.ox, re.oy, re.oz, re.t_max);
                                stage[0][1][tid] = make_float4(re.dx, re.dy, re.dz, re.time);
                            }
                        }
                        cur.dim = dim + c;
                        dim = cur.dim;
                        PHC_END(8);
                    }
                }
            }
        }
        const bool still_live = active && (flags & (F_EXT | F_PSH | F_PMIS)) != 0;

        // ---- block-aggregated queue appends --------------------------------------------------------------------------------------------
        PHC_BEGIN(9);
        const uint64_t m_ext = __ballot(want_ext), m_mis = __ballot(want_mis), m_sh = __ballot(want_sh), m_lv = __ballot(still_live);
        if (lane == 0) {
            wave_cnt[0][wid] = (uint32_t)(__popcll(m_ext) + __popcll(m_mis));
            wave_cnt[1][wid] = (uint32_t)__popcll(m_sh);
            wave_cnt[2][wid] = (uint32_t)__popcll(m_lv);
        }
        __syncthreads();
        if (tid < 3) {
            uint32_t tot = 0;
            for (uint32_t k = 0; k < PH_SHADE_BLOCK / 64; k++) tot += wave_cnt[tid][k];
            uint32_t* ctr = tid == 0 ? &next->n_cl : (tid == 1 ? &next->n_sh : &next->n_live);
            q_base[tid] = tot ? atomicAdd(ctr, tot) : 0u;
        }
        __syncthreads();
        uint32_t cl_slot = q_base[0], sh_slot = q_base[1], lv_slot = q_base[2];
        for (uint32_t k = 0; k < wid; k++) { cl_slot += wave_cnt[0][k]; sh_slot += wave_cnt[1][k]; lv_slot += wave_cnt[2][k]; }
        cl_slot += (uint32_t)(__popcll(m_ext & lane_lt) + __popcll(m_mis & lane_lt));
        sh_slot += (uint32_t)__popcll(m_sh & lane_lt);
        lv_slot += (uint32_t)__popcll(m_lv & lane_lt);

        if (active) {
            uint32_t ext_slot = 0, mis_slot = 0;
            if (want_ext) {
                ext_slot = cl_slot;
                float4* d = reinterpret_cast<float4*>(rays_out + ext_slot);
                const float4 ro = stage[0][0][tid], rdv = stage[0][1][tid];
                d[0] = ro; d[1] = rdv;
                if (w.sort_grid.mode) w.keys_cl[ext_slot] = ray_sort_key(w.sort_grid, ro.x, ro.y, ro.z, rdv.x, rdv.y, rdv.z);
            }
            if (want_mis) {
                mis_slot = cl_slot + (want_ext ? 1u : 0u);
                float4* d = reinterpret_cast<float4*>(rays_out + mis_slot);
                const float4 ro = stage[1][0][tid], rdv = stage[1][1][tid];
                d[0] = ro; d[1] = rdv;
                if (w.sort_grid.mode) w.keys_cl[mis_slot] = ray_sort_key(w.sort_grid, ro.x, ro.y, ro.z, rdv.x, rdv.y, rdv.z);
            }
            if (want_sh) {
                float4* d = reinterpret_cast<float4*>(w.rays_sh + sh_slot);
                const float4 ro = stage[2][0][tid], rdv = stage[2][1][tid];
                d[0] = ro; d[1] = rdv;
                if (w.sort_grid.mode) w.keys_sh[sh_slot] = ray_sort_key(w.sort_grid, ro.x, ro.y, ro.z, rdv.x, rdv.y, rdv.z);
            }
            if (still_live) {
                live_out[lv_slot] = pid;
                w.s_idx[(it + 1) & 1][lv_slot] = make_uint4(ext_slot, mis_slot, sh_slot, flags | (bounces << 8) | (dim << 16));
                w.s_L[(it + 1) & 1][lv_slot] = make_float4(L.r, L.g, L.b, pick_pdf);
                w.s_beta[(it + 1) & 1][lv_slot] = make_float4(beta.r, beta.g, beta.b, eta_scale);
                w.s_prev[(it + 1) & 1][lv_slot] = i;
            } else {
                // path finished: radiance sanitising of render_tile (sampler_integrator.rs:373-397)
                if (has_nans(L)) L = mks1(0.0f);
                else if (lum_y(L) < -1e-5f) L = mks1(0.0f);
                else if (__builtin_isinf(lum_y(L))) L = mks1(0.0f);
                const uint32_t ppix = pid / w.chunk_spp;
                const size_t gsi = (size_t)(w.s0 + (pid - ppix * w.chunk_spp)) * w.n_px + ppix;
                float4 rec = w.rec_L[gsi];
                rec.x = L.r; rec.y = L.g; rec.z = L.b;
                w.rec_L[gsi] = rec;
            }
        }
        __syncthreads();  // stage / wave_cnt are reused by the next grid-stride round
        PHC_END(9);
    }
    PHC_END(10);
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
    if (threadIdx.x < 3 * PHC_N && w.phase) atomicAdd(w.phase + threadIdx.x, phc_lds[threadIdx.x]);
#endif
    for (int o = 32; o > 0; o >>= 1) { n_paths_total += __shfl_xor(n_paths_total, o); n_paths_zero += __shfl_xor(n_paths_zero, o); }
    if (lane == 0) { atomicAdd(&blk_stats[0], n_paths_total); atomicAdd(&blk_stats[1], n_paths_zero); }
    __syncthreads();
    if (tid == 0 && blk_stats[0]) atomicAdd(&w.stats->paths_total, (unsigned long long)blk_stats[0]);
    if (tid == 1 && blk_stats[1]) atomicAdd(&w.stats->paths_zero, (unsigned long long)blk_stats[1]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// K7a: FilmTile::add_sample (core/src/film/film_tile.rs:62-108), one thread per (tile, tile pixel).  The thread walks the
// tile's samples that can reach its pixel in the reference's order (pixels row-major, then sample index) and accumulates
// exactly the terms the reference adds, in the same order.
struct FilmParams {
    FilmRec film;
    const TileInfo* tiles; uint32_t n_tiles;
    uint32_t slot_w, slot_h;      // per-tile slot in the tile buffer: slot_w*slot_h float4 {contrib rgb, weight sum}
    uint32_t spp, n_px;
    const float4* rec_L; const float* rec_py;   // [sample][pixel of the rank's pixel list]
    const uint8_t* px_rounded;                  // pixels with a sample that rounded up onto the next pixel's coordinate (raygen_kernel)
    float4* tile_buf;
};
__global__ __launch_bounds__(256) void film_tiles_kernel(FilmParams p) {
    const uint32_t slot_px = p.slot_w * p.slot_h;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)p.n_tiles * slot_px) return;
    const uint32_t lt = (uint32_t)(gid / slot_px), k = (uint32_t)(gid % slot_px);
    const TileInfo t = p.tiles[lt];
    const int pw = t.pb[2] - t.pb[0], phh = t.pb[3] - t.pb[1];
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int kx = (int)(k % p.slot_w), ky = (int)(k / p.slot_w);
    if (kx < pw && ky < phh) {
        const int x = t.pb[0] + kx, y = t.pb[1] + ky;
        const float rx = p.film.radius[0], ry = p.film.radius[1];
        // a sample at film position p reaches pixel x iff ceil(p-.5-r) <= x <= floor(p-.5+r), i.e. p in [x+.5-r, x+.5+r]; samples of
        // pixel sx have p in [sx, sx+1), so only sx in [floor(x+.5-r), floor(x+.5+r)] can contribute (evaluated in f64: exact) ...
        const int sx0n = (int)floor((double)x + 0.5 - (double)rx), sy0n = (int)floor((double)y + 0.5 - (double)ry);
        // ... except that `pixel + offset` is an f32 sum: an offset within half an ulp of 1 puts the sample ON the next pixel's coordinate, p = sx + 1 (beyond x = 512
        // every offset above 1 - 2^-15 does; at 512 spp the Halton points reach 1 - 2^-17).  Where x + .5 - r is a whole number such a sample of pixel sx0n - 1 still
        // reaches x: that column / row is scanned too, for the samples whose position is exactly sx + 1 (round 3: found by the configs[4] crop at 512 spp)
        const int sx0e = (int)ceil((double)x - 0.5 - (double)rx) < sx0n ? sx0n - 1 : sx0n, sy0e = (int)ceil((double)y - 0.5 - (double)ry) < sy0n ? sy0n - 1 : sy0n;
        const int sx0 = pmaxi(sx0e, t.tb[0]), sx1 = pmini((int)floor((double)x + 0.5 + (double)rx), t.tb[2] - 1);
        const int sy0 = pmaxi(sy0e, t.tb[1]), sy1 = pmini((int)floor((double)y + 0.5 + (double)ry), t.tb[3] - 1);
        const int tw = t.tb[2] - t.tb[0];
        for (int sy = sy0; sy <= sy1; sy++)
            for (int sx = sx0; sx <= sx1; sx++) {
                const size_t pix = (size_t)t.px_off + (size_t)(sy - t.tb[1]) * tw + (size_t)(sx - t.tb[0]);
                const bool up_x = sx < sx0n, up_y = sy < sy0n;   // only samples rounded up onto the next column / row matter from here ...
                if ((up_x || up_y) && !p.px_rounded[pix]) continue;   // ... and raygen noted which pixels have any (a few per cent of them at 512 spp)
                for (uint32_t s = 0; s < p.spp; s++) {
                    const float4 r = p.rec_L[(size_t)s * p.n_px + pix];   // (both loads issued before either is tested: one latency per sample, not two)
                    const float pfy = p.rec_py[(size_t)s * p.n_px + pix];
                    if (r.w != r.w) continue;  // pixel outside pixel_bounds: no sample was taken
                    if ((up_y && pfy != (float)(sy + 1)) || (up_x && r.w != (float)(sx + 1))) continue;
                    const float pfx = r.w;
                    spec l = mks(r.x, r.y, r.z);
                    const float ly = lum_y(l);
                    if (ly > p.film.max_lum) l = l * p.film.max_lum / ly;
                    const float pdx = pfx - 0.5f, pdy = pfy - 0.5f;
                    int p0x = f2i_sat(ceilf(pdx - rx)), p0y = f2i_sat(ceilf(pdy - ry));
                    int p1x = f2i_sat(floorf(pdx + rx)) + 1, p1y = f2i_sat(floorf(pdy + ry)) + 1;
                    p0x = pmaxi(p0x, t.pb[0]); p0y = pmaxi(p0y, t.pb[1]); p1x = pmini(p1x, t.pb[2]); p1y = pmini(p1y, t.pb[3]);
                    if (x < p0x || x >= p1x || y < p0y || y >= p1y) continue;
                    const float fx = pabs(((float)x - pdx) * p.film.inv_radius[0] * 16.0f);
                    const float fy = pabs(((float)y - pdy) * p.film.inv_radius[1] * 16.0f);
                    const uint32_t ix = f2u_sat(pminf(floorf(fx), 15.0f)), iy = f2u_sat(pminf(floorf(fy), 15.0f));
                    const float fw = p.film.table[iy * 16 + ix];
                    const spec c = l * 1.0f * fw;  // l * sample_weight * filter_weight (ray weight is 1 for this camera)
                    acc.x += c.r; acc.y += c.g; acc.z += c.b; acc.w += fw;
                }
            }
    }
    p.tile_buf[(size_t)lt * slot_px + k] = acc;
}

// K7b: Film::merge_film_tile (core/src/film/mod.rs:220-279) over all tiles in increasing tile index, one thread per film pixel
struct MergeParams {
    FilmRec film;
    int32_t sb[4]; int32_t tile_size, ntx, nty, parts;
    uint32_t slot_w, slot_h;
    const float4* bufs[PH_MAX_TILE_PARTS];
    float* out_xyz; float* out_w;
};
PH_DEV void tile_pixel_bounds(const FilmRec& f, const int tb[4], int pb[4]) {  // Film::get_film_tile (film/mod.rs:182-198)
    const int p0x = f2i_sat(ceilf((float)tb[0] - 0.5f - f.radius[0])), p0y = f2i_sat(ceilf((float)tb[1] - 0.5f - f.radius[1]));
    const int p1x = f2i_sat(floorf((float)tb[2] - 0.5f + f.radius[0])) + 1, p1y = f2i_sat(floorf((float)tb[3] - 0.5f + f.radius[1])) + 1;
    pb[0] = pmaxi(p0x, f.crop[0]); pb[1] = pmaxi(p0y, f.crop[1]); pb[2] = pmini(p1x, f.crop[2]); pb[3] = pmini(p1y, f.crop[3]);
}
__global__ __launch_bounds__(256) void merge_kernel(MergeParams p) {
    const int cw = p.film.crop[2] - p.film.crop[0], ch = p.film.crop[3] - p.film.crop[1];
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cw <= 0 || ch <= 0 || gid >= (uint64_t)cw * ch) return;
    const int x = p.film.crop[0] + (int)(gid % cw), y = p.film.crop[1] + (int)(gid / cw);
    const int ext_x = (int)ceilf(p.film.radius[0] + 0.5f) + 1, ext_y = (int)ceilf(p.film.radius[1] + 0.5f) + 1;
    const int tx_lo = pmaxi((x - ext_x - p.sb[0]) / p.tile_size - 1, 0), tx_hi = pmini((x + ext_x - p.sb[0]) / p.tile_size + 1, p.ntx - 1);
    const int ty_lo = pmaxi((y - ext_y - p.sb[1]) / p.tile_size - 1, 0), ty_hi = pmini((y + ext_y - p.sb[1]) / p.tile_size + 1, p.nty - 1);
    float X = 0.0f, Y = 0.0f, Z = 0.0f, W = 0.0f;
    const uint32_t slot_px = p.slot_w * p.slot_h;
    for (int ty = ty_lo; ty <= ty_hi; ty++)
        for (int tx = tx_lo; tx <= tx_hi; tx++) {
            int tb[4], pb[4];
            tb[0] = p.sb[0] + tx * p.tile_size; tb[2] = pmini(tb[0] + p.tile_size, p.sb[2]);
            tb[1] = p.sb[1] + ty * p.tile_size; tb[3] = pmini(tb[1] + p.tile_size, p.sb[3]);
            tile_pixel_bounds(p.film, tb, pb);
            if (x < pb[0] || x >= pb[2] || y < pb[1] || y >= pb[3]) continue;
            const uint32_t t = (uint32_t)(ty * p.ntx + tx);
            const float4 c = p.bufs[t % p.parts][(size_t)(t / p.parts) * slot_px + (size_t)(y - pb[1]) * p.slot_w + (size_t)(x - pb[0])];
            // contrib_sum.to_xyz() (spectrum/common.rs:349-355)
            X += 0.412453f * c.x + 0.357580f * c.y + 0.180423f * c.z;
            Y += 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z;
            Z += 0.019334f * c.x + 0.119193f * c.y + 0.950227f * c.z;
            W += c.w;
        }
    p.out_xyz[3 * gid] = X; p.out_xyz[3 * gid + 1] = Y; p.out_xyz[3 * gid + 2] = Z; p.out_w[gid] = W;
}

__global__ void preset_counters_kernel(IterCounters* ctr, DevStats* stats, uint32_t n) {
    ctr[0].n_cl = n; ctr[0].n_live = n;
    stats->camera_rays += n;
}

// camera rays only (parity harness for the sampler + camera rows of SURVEY §8a)
__global__ void camera_rays_kernel(DeviceScene sc, CameraRec cam, SamplerRec sp, int x0, int y0, int x1, int y1, uint32_t s, RayIn* out, float* out_pf) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wdt = x1 - x0, hgt = y1 - y0;
    if (wdt <= 0 || hgt <= 0 || gid >= (uint32_t)(wdt * hgt)) return;
    const int x = x0 + (int)(gid % wdt), y = y0 + (int)(gid / wdt);
    SamplerCursor c = cursor_for(sc, sp, x, y, s, 0);
    f2 fs = get_2d(sc, sp, c);
    f2 pf = mk2((float)x + fs.x, (float)y + fs.y);
    float tm = get_1d(sc, sp, c);
    f2 lens = get_2d(sc, sp, c);
    RayIn r;
    generate_camera_ray(cam, pf, tm, lens, r);
    store_ray(out + gid, r);
    if (out_pf) { out_pf[2 * gid] = pf.x; out_pf[2 * gid + 1] = pf.y; }
}

}  // namespace ph

// =================================================================================================================================
// host side
// =================================================================================================================================
struct Wavefront {
    // geometry of the tile decomposition for the current (tile_size, part, parts, film, sampler)
    std::vector<ph::TileInfo> tiles;
    std::vector<int2> px_xy;
    int sb[4] = {0, 0, 0, 0}, ntx = 0, nty = 0, tile_size = 0, part = 0, parts = 0;
    uint32_t slot_w = 0, slot_h = 0;
    int tiles_key[12] = {0}; bool tiles_valid = false;  // what the lists above (and their device copies) were built for
    DevBuf d_tiles, d_px, d_rays_cl[2], d_hits, d_rays_sh, d_occ, d_live[2], d_ctr, d_stats, d_cam, d_tex_out;
    DevBuf d_sL, d_sbeta, d_sA, d_sf2, d_sbold, d_sidx, d_sprev, d_recL, d_recpy, d_rounded, d_tilebuf, d_xyz, d_w;
    DevBuf d_vox_slot, d_sp_pool, d_sp_list, d_sp_ctr, d_sp_halton;  // SpatialLightDistribution tables (spatial.h)
    DevBuf d_order, d_sort_bins, d_heads, d_keys_cl, d_keys_sh;  // ray binning between rounds (raysort.h), per-XCD queue heads
    DevBuf d_morder, d_mkeys, d_mbins;  // shade-side work queues (matsort.h)
    size_t n_vox = 0;   // voxels of the last spatial render's table (d_vox_slot)
    std::vector<hipEvent_t> events;
};

namespace phost {

void free_wavefront(PbrtHipScene* s) {
    Wavefront* w = s->wf;
    if (!w) return;
    for (DevBuf* b : {&w->d_tiles, &w->d_px, &w->d_rays_cl[0], &w->d_rays_cl[1], &w->d_hits, &w->d_rays_sh, &w->d_occ, &w->d_live[0], &w->d_live[1], &w->d_ctr,
                      &w->d_stats, &w->d_cam, &w->d_tex_out, &w->d_sL, &w->d_sbeta, &w->d_sA, &w->d_sf2, &w->d_sbold, &w->d_sidx, &w->d_sprev, &w->d_rounded, &w->d_recL, &w->d_recpy, &w->d_tilebuf, &w->d_xyz, &w->d_w,
                      &w->d_vox_slot, &w->d_sp_pool, &w->d_sp_list, &w->d_sp_ctr, &w->d_sp_halton, &w->d_order, &w->d_sort_bins, &w->d_heads, &w->d_keys_cl, &w->d_keys_sh, &w->d_morder, &w->d_mkeys, &w->d_mbins})
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : w->events) (void)hipEventDestroy(e);
    delete w;
    s->wf = nullptr;
}

static int sat_i(float v) { if (v != v) return 0; if (v >= 2147483648.0f) return 2147483647; if (v <= -2147483648.0f) return (int)0x80000000; return (int)v; }

// Frame-wide tile grid: Film::get_sample_bounds (film/mod.rs:150-159), the tile counts of SamplerIntegrator::render (sampler_integrator.rs:252-259)
// and the tile-buffer slot big enough for any tile's FilmTile pixel bounds.  Independent of which part of the tiles a rank renders.
struct TileGrid { int sb[4]; int ntx, nty; uint32_t slot_w, slot_h; };
static TileGrid tile_grid(const FilmRec& f, int tile_size) {
    TileGrid g;
    g.sb[0] = sat_i(std::floor((float)f.crop[0] + 0.5f - f.radius[0])); g.sb[1] = sat_i(std::floor((float)f.crop[1] + 0.5f - f.radius[1]));
    g.sb[2] = sat_i(std::ceil((float)f.crop[2] - 0.5f + f.radius[0])); g.sb[3] = sat_i(std::ceil((float)f.crop[3] - 0.5f + f.radius[1]));
    g.ntx = std::max((g.sb[2] - g.sb[0] + tile_size - 1) / tile_size, 0); g.nty = std::max((g.sb[3] - g.sb[1] + tile_size - 1) / tile_size, 0);
    const int e_lo_x = sat_i(std::floor(0.5f + f.radius[0])), e_hi_x = sat_i(std::floor(f.radius[0] - 0.5f)) + 1;
    const int e_lo_y = sat_i(std::floor(0.5f + f.radius[1])), e_hi_y = sat_i(std::floor(f.radius[1] - 0.5f)) + 1;
    g.slot_w = (uint32_t)std::max(tile_size + e_lo_x + e_hi_x, 1); g.slot_h = (uint32_t)std::max(tile_size + e_lo_y + e_hi_y, 1);
    return g;
}

// Tile decomposition exactly as SamplerIntegrator::render / render_tile enumerate it (sampler_integrator.rs:252-259, 314-336).
// The tile and pixel lists of a rank depend only on (film window, filter radius, tile size, part, parts): they are kept on the device
// between frames and rebuilt only when that key changes.
static int setup_tiles(PbrtHipScene* s, int tile_size, int part, int parts) {
    if (!s->wf) s->wf = new Wavefront();
    Wavefront& w = *s->wf;
    const FilmRec& f = s->film;
    const TileGrid g = tile_grid(f, tile_size);
    int key[12] = {f.crop[0], f.crop[1], f.crop[2], f.crop[3], 0, 0, tile_size, part, parts, g.ntx, g.nty, 1};
    std::memcpy(&key[4], &f.radius[0], 4); std::memcpy(&key[5], &f.radius[1], 4);
    if (w.tiles_valid && std::memcmp(key, w.tiles_key, sizeof key) == 0) return PBRT_HIP_OK;
    w.tiles_valid = false;
    const int* sb = g.sb; const int ntx = g.ntx, nty = g.nty;
    w.tiles.clear(); w.px_xy.clear();
    std::memcpy(w.sb, sb, sizeof(w.sb)); w.ntx = ntx; w.nty = nty; w.tile_size = tile_size; w.part = part; w.parts = parts;
    w.slot_w = g.slot_w; w.slot_h = g.slot_h;
    for (int t = part; t < ntx * nty; t += parts) {
        ph::TileInfo ti{};
        const int tx = t % ntx, ty = t / ntx;
        ti.tb[0] = sb[0] + tx * tile_size; ti.tb[2] = std::min(ti.tb[0] + tile_size, sb[2]);
        ti.tb[1] = sb[1] + ty * tile_size; ti.tb[3] = std::min(ti.tb[1] + tile_size, sb[3]);
        const int p0x = sat_i(std::ceil((float)ti.tb[0] - 0.5f - f.radius[0])), p0y = sat_i(std::ceil((float)ti.tb[1] - 0.5f - f.radius[1]));
        const int p1x = sat_i(std::floor((float)ti.tb[2] - 0.5f + f.radius[0])) + 1, p1y = sat_i(std::floor((float)ti.tb[3] - 0.5f + f.radius[1])) + 1;
        ti.pb[0] = std::max(p0x, f.crop[0]); ti.pb[1] = std::max(p0y, f.crop[1]); ti.pb[2] = std::min(p1x, f.crop[2]); ti.pb[3] = std::min(p1y, f.crop[3]);
        ti.px_off = (uint32_t)w.px_xy.size(); ti.tile_index = (uint32_t)t;
        for (int y = ti.tb[1]; y < ti.tb[3]; y++)
            for (int x = ti.tb[0]; x < ti.tb[2]; x++) w.px_xy.push_back(make_int2(x, y));  // Bounds2i iteration order (bounds2.rs:347-359)
        w.tiles.push_back(ti);
    }
    int rc;
    if ((rc = ensure_buf(s, w.d_tiles, std::max<size_t>(w.tiles.size(), 1) * sizeof(ph::TileInfo)))) return rc;
    if ((rc = ensure_buf(s, w.d_px, std::max<size_t>(w.px_xy.size(), 1) * sizeof(int2)))) return rc;
    if (!w.tiles.empty()) PH_CHECK(s, hipMemcpy(w.d_tiles.p, w.tiles.data(), w.tiles.size() * sizeof(ph::TileInfo), hipMemcpyHostToDevice));
    if (!w.px_xy.empty()) PH_CHECK(s, hipMemcpy(w.d_px.p, w.px_xy.data(), w.px_xy.size() * sizeof(int2), hipMemcpyHostToDevice));
    std::memcpy(w.tiles_key, key, sizeof key); w.tiles_valid = true;
    return PBRT_HIP_OK;
}

size_t tile_buffer_floats_for(const PbrtHipScene* s, int tile_size, int part, int parts) {
    const TileGrid g = tile_grid(s->film, tile_size);
    const int n = g.ntx * g.nty;
    const size_t local = n > part ? (size_t)(n - part + parts - 1) / parts : 0;
    return std::max<size_t>(local, 1) * (size_t)g.slot_w * g.slot_h * 4;
}

static hipEvent_t get_event(PbrtHipScene* s, size_t i) {
    Wavefront& w = *s->wf;
    while (w.events.size() <= i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; w.events.push_back(e); }
    return w.events[i];
}

int check_render_args(PbrtHipScene* s, int max_depth, int light_strategy, const int* pixel_bounds, int tile_size, int part, int parts) {
    if (!s || !pixel_bounds) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: null argument");
    if (!s->built || !s->have_camera || !s->have_film || !s->have_sampler)
        return set_err(s, PBRT_HIP_ERR_STATE, "render: camera, film, sampler and build_accel must be set first");
    if (tile_size <= 0 || parts <= 0 || parts > PH_MAX_TILE_PARTS || part < 0 || part >= parts) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: bad tile partition");
    if (max_depth < 0 || max_depth > 200) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: max_depth out of range");
    if (light_strategy < 0 || light_strategy > 2) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: bad light strategy");
    if (s->sampler.kind == 1 && s->sobol32.empty()) return set_err(s, PBRT_HIP_ERR_STATE, "render: sobol tables not set (pbrt_hip_set_sobol_tables)");
    // sampler dimension budget: 5 + per bounce (1+2+2) + 2 + 1; HaltonSampler asserts dim <= 1000 (halton.rs:106-110)
    if (5 + 8 * (max_depth + 1) >= (s->sampler.kind == 0 ? 1000 : 1024)) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "render: path would exceed the sampler's dimension table");
    return PBRT_HIP_OK;
}

// SpatialLightDistribution::new (spatial.rs:57-88): voxel resolution from the scene bounds, tables reset for this render
// (the reference builds the distribution in Integrator::preprocess and fills it lazily while rendering, so the cost of the
// voxel distributions is inside the timed render here too).
static int setup_spatial(PbrtHipScene* s, ph::SpatialRec& sr) {
    Wavefront& w = *s->wf;
    const size_t n_lights = s->lights.size();
    float diag[3];
    for (int i = 0; i < 3; i++) { sr.lo[i] = s->bvh.root_lo[i]; sr.hi[i] = s->bvh.root_hi[i]; diag[i] = sr.hi[i] - sr.lo[i]; }
    const int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);  // Bounds3::maximum_extent
    const float bmax = diag[ext];
    size_t nvox = 1;
    for (int i = 0; i < 3; i++) {
        const float r = std::round(diag[i] / bmax * 64.0f);  // f32::round: half away from zero; `as usize` saturates, NaN -> 0
        long long v = (r != r) ? 0 : (r <= 0.0f ? 0 : (r >= 1048575.0f ? 1048575 : (long long)r));
        sr.nv[i] = (int32_t)std::max<long long>(1, v);
        nvox *= (size_t)sr.nv[i];
    }
    sr.stride = (uint32_t)(2 * n_lights + 2);
    size_t free_b = 0, total_b = 0;
    PH_CHECK(s, hipMemGetInfo(&free_b, &total_b));
    const size_t slot_bytes = (size_t)sr.stride * 4;
    size_t budget = free_b / 2 + w.d_sp_pool.bytes;  // the pool may take half of what is free (plus what it already holds)
    if (const char* e = std::getenv("PBRT_HIP_SPATIAL_POOL_BYTES")) { long long v = std::atoll(e); if (v > 0) budget = (size_t)v; }
    const size_t cap = std::min(nvox, budget / slot_bytes);
    if (cap == 0) return set_err(s, PBRT_HIP_ERR_OOM, "render: no memory for the SpatialLightDistribution pool");
    sr.capacity = (uint32_t)cap;
    int rc;
    if ((rc = ensure_buf(s, w.d_vox_slot, nvox * 4))) return rc;
    w.n_vox = nvox;
    if ((rc = ensure_buf(s, w.d_sp_list, cap * 4))) return rc;
    if ((rc = ensure_buf(s, w.d_sp_ctr, 64))) return rc;
    if ((rc = ensure_buf(s, w.d_sp_pool, cap * slot_bytes))) return rc;
    if (!w.d_sp_halton.p) {
        if ((rc = ensure_buf(s, w.d_sp_halton, 128 * 5 * 4))) return rc;
        static const uint32_t bases[5] = {2, 3, 5, 7, 11};
        float h[128 * 5];
        for (uint64_t i = 0; i < 128; i++)
            for (int d = 0; d < 5; d++) h[5 * i + d] = hm::radical_inverse(bases[d], i);
        PH_CHECK(s, hipMemcpy(w.d_sp_halton.p, h, sizeof h, hipMemcpyHostToDevice));
    }
    PH_CHECK(s, hipMemsetAsync(w.d_vox_slot.p, 0xFF, nvox * 4, s->stream));
    PH_CHECK(s, hipMemsetAsync(w.d_sp_ctr.p, 0, 64, s->stream));
    sr.enabled = 1;
    sr.vox_slot = (int32_t*)w.d_vox_slot.p; sr.pool = (float*)w.d_sp_pool.p; sr.new_list = (uint32_t*)w.d_sp_list.p;
    sr.counters = (uint32_t*)w.d_sp_ctr.p; sr.halton = (const float*)w.d_sp_halton.p;
    return PBRT_HIP_OK;
}

// renders this rank's tiles into d_tile_buffer (device)
int render_tiles(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size, int part,
                        int parts, void* d_tile_buffer, PbrtHipStats* out_stats) {
    int rc;
    PH_CHECK(s, hipSetDevice(s->device));
    if ((rc = upload_scene(s))) return rc;
    const bool spatial = light_strategy == 2 && s->lights.size() > 1;  // one light -> uniform (light_distrib/mod.rs:59-64)
    if ((rc = upload_light_distribution(s, light_strategy == 2 ? 0 : light_strategy))) return rc;
    if ((rc = setup_tiles(s, tile_size, part, parts))) return rc;
    Wavefront& w = *s->wf;
    const uint32_t n_px = (uint32_t)w.px_xy.size();
    const uint32_t spp = s->sampler.spp;
    if (out_stats) std::memset(out_stats, 0, sizeof(*out_stats));
    const size_t tile_floats = tile_buffer_floats_for(s, tile_size, part, parts);
    if (n_px == 0) { PH_CHECK(s, hipMemsetAsync(d_tile_buffer, 0, tile_floats * 4, s->stream)); PH_CHECK(s, hipStreamSynchronize(s->stream)); return PBRT_HIP_OK; }
    if ((uint64_t)n_px * spp >= (1ull << 40)) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "render: too many samples");

    // ---- chunking: B = n_px * chunk_spp paths in flight ----------------------------------------------------------------------
    // 128 Mi paths per chunk where the card has room for them (353 B of queues and path state per path, 481 B with a texture pass: 47 – 65 GB of the MI355X's 288 GB):
    // large chunks bin better (more rays per origin cell and round) and pay fewer launch tails — configs[2] 913 -> 878 ms per frame, configs[3] 934 -> 898 ms against the
    // 32 Mi of round 1 (gpurun r02aa; 256 Mi, the whole frame at once, adds nothing: 875 / 899 ms).  At most 30 % of the device's memory goes to one chunk.
    const size_t per_path = 2 * 2 * sizeof(ph::RayIn) + 2 * sizeof(ph::HitOut) + sizeof(ph::RayIn) + 1 + 2 * 4 + 6 * 2 * 16 + 2 * 4   // ray / hit queues, live lists, path state (two buffers)
                            + 3 * 4 + 3 * 4 + (s->textured_materials ? sizeof(TexOut) : 0)                                                // + bin keys and order + the texture pass's records
                            + 4 + 2;                                                                                                       // + the shade-side work queues' order and keys
    // everything in this context that grows with the chunk, as allocated now: a chunk may reuse it
    auto chunk_bufs = [&]() { return std::vector<DevBuf*>{&w.d_rays_cl[0], &w.d_rays_cl[1], &w.d_hits, &w.d_rays_sh, &w.d_occ, &w.d_live[0], &w.d_live[1], &w.d_sL, &w.d_sbeta, &w.d_sA, &w.d_sf2,
                                                         &w.d_sbold, &w.d_sidx, &w.d_sprev, &w.d_order, &w.d_keys_cl, &w.d_keys_sh, &w.d_tex_out, &w.d_morder, &w.d_mkeys}; };
    size_t max_paths = 128u << 20;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
            size_t held = 0;
            for (DevBuf* b : chunk_bufs()) held += b->bytes;
            const size_t rec_need = (size_t)n_px * spp * 20, rec_have = w.d_recL.bytes + w.d_recpy.bytes;
            size_t avail = free_b / 10 * 8 + held;   // what this context holds already counts as available to it; other contexts on the card (repeated-ordinal handles, other ranks) keep theirs
            avail = avail > (rec_need > rec_have ? rec_need - rec_have : 0) ? avail - (rec_need > rec_have ? rec_need - rec_have : 0) : 0;
            max_paths = std::max<size_t>(1u << 20, std::min<size_t>(max_paths, std::min(total_b / 10 * 3, avail) / per_path));
        }
    }
    if (const char* e = std::getenv("PBRT_HIP_MAX_PATHS")) { long long v = std::atoll(e); if (v > 0) max_paths = (size_t)v; }
    uint32_t chunk_spp = (uint32_t)std::max<size_t>(1, std::min<size_t>(spp, max_paths / std::max<uint32_t>(n_px, 1)));
    const int n_iter = max_depth + 1;
    // Material "none" surfaces are passed through without counting a bounce, so a path may need more rounds than max_depth + 1: those are
    // run one at a time while paths remain (host reads the live count), up to kMaxNullSkips more.
    const int kMaxNullSkips = 1024;
    const int n_iter_cap = s->has_none_material ? n_iter + kMaxNullSkips : n_iter;
    static const int sort_mode = []() { const char* e = std::getenv("PBRT_HIP_SORT_RAYS"); int v = e ? std::atoi(e) : 1; return (v < 0 || v > 3) ? 1 : v; }();
    // shade-side work queues (matsort.h): scenes with anything but constant matte — the general-BSDF kernel's branches and the texture pass's programs depend on the material.
    // PBRT_HIP_MATERIAL_QUEUES=0 walks the list in queue order as rounds 1 - 3 did (A/B aid; same film)
    static const bool mq_env = []() { const char* e = std::getenv("PBRT_HIP_MATERIAL_QUEUES"); return !(e && std::atoi(e) == 0); }();
    const bool mat_queues = mq_env && (s->general_materials || s->textured_materials);
    if ((rc = ensure_buf(s, w.d_ctr, (size_t)(n_iter_cap + 2) * sizeof(ph::IterCounters)))) return rc;
    if ((rc = ensure_buf(s, w.d_stats, 64 + 6 * PHC_N * 8))) return rc;   // DevStats (+ the shade kernel's phase tallies in measurement builds)
    if ((rc = ensure_buf(s, w.d_recL, (size_t)n_px * spp * 16))) return rc;
    if ((rc = ensure_buf(s, w.d_recpy, (size_t)n_px * spp * 4))) return rc;
    if ((rc = ensure_buf(s, w.d_rounded, (size_t)n_px))) return rc;
    PH_CHECK(s, hipMemsetAsync(w.d_rounded.p, 0, (size_t)n_px, s->stream));
    auto alloc_chunk = [&](size_t Bc) -> int {
        int r;
        if ((r = ensure_buf(s, w.d_rays_cl[0], 2 * Bc * sizeof(ph::RayIn)))) return r;
        if ((r = ensure_buf(s, w.d_rays_cl[1], 2 * Bc * sizeof(ph::RayIn)))) return r;
        if ((r = ensure_buf(s, w.d_hits, 2 * Bc * sizeof(ph::HitOut)))) return r;
        if ((r = ensure_buf(s, w.d_rays_sh, Bc * sizeof(ph::RayIn)))) return r;
        if ((r = ensure_buf(s, w.d_occ, Bc))) return r;
        if ((r = ensure_buf(s, w.d_live[0], Bc * 4))) return r;
        if ((r = ensure_buf(s, w.d_live[1], Bc * 4))) return r;
        for (DevBuf* b : {&w.d_sL, &w.d_sbeta, &w.d_sA, &w.d_sf2, &w.d_sbold, &w.d_sidx})
            if ((r = ensure_buf(s, *b, 2 * Bc * 16))) return r;   // two buffers each: queue-ordered, read from one and written to the other (WfParams)
        if ((r = ensure_buf(s, w.d_sprev, 2 * Bc * 4))) return r;
        if (sort_mode) {
            if ((r = ensure_buf(s, w.d_order, 3 * Bc * 4))) return r;
            if ((r = ensure_buf(s, w.d_keys_cl, 2 * Bc * 4))) return r;
            if ((r = ensure_buf(s, w.d_keys_sh, Bc * 4))) return r;
        }
        if (s->textured_materials && (r = ensure_buf(s, w.d_tex_out, Bc * sizeof(TexOut)))) return r;
        if (mat_queues) {
            if ((r = ensure_buf(s, w.d_morder, Bc * 4))) return r;
            if ((r = ensure_buf(s, w.d_mkeys, Bc * 2))) return r;
        }
        return PBRT_HIP_OK;
    };
    // the estimate above can be wrong (fragmentation, another context allocating meanwhile): on hipErrorOutOfMemory the chunk is halved and tried again before the call gives up
    // (test hook: PBRT_HIP_TEST_CHUNK_OOM=k makes the first k attempts of every render call fail as an out-of-memory allocation would)
    int forced_oom = 0;
    if (const char* e = std::getenv("PBRT_HIP_TEST_CHUNK_OOM")) forced_oom = std::max(0, std::atoi(e));
    bool retried = false;
    for (;;) {
        if ((size_t)n_px * chunk_spp >= 0x7FFF0000ull) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "render: tile range too large for one rank; use more tile_parts");
        if (forced_oom > 0) { forced_oom--; rc = set_err(s, PBRT_HIP_ERR_OOM, "render: out of device memory (forced by PBRT_HIP_TEST_CHUNK_OOM)"); }
        else rc = alloc_chunk((size_t)n_px * chunk_spp);
        if (rc == PBRT_HIP_OK) { if (retried) s->err.clear(); break; }   // a retry that succeeded leaves no error text behind
        retried = true;
        if (rc != PBRT_HIP_ERR_OOM || chunk_spp == 1) return rc;
        for (DevBuf* b : chunk_bufs()) if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->bytes = 0; }
        chunk_spp = (chunk_spp + 1) / 2;
    }
    const size_t B = (size_t)n_px * chunk_spp;

    if ((rc = ensure_traversal_workspace(s))) return rc;
    // ray binning between rounds and per-XCD queue heads (raysort.h, traverse.h)
    static const int n_heads = []() { const char* e = std::getenv("PBRT_HIP_TRAV_HEADS"); int v = e ? std::atoi(e) : 8; return (v < 1 || v > 8) ? 8 : v; }();
    static const int sort_blocks = []() { const char* e = std::getenv("PBRT_HIP_SORT_BLOCKS"); int v = e ? std::atoi(e) : 1024; return (v < 64 || v > 8192) ? 1024 : v; }();
    static const int head_chunk = []() { const char* e = std::getenv("PBRT_HIP_HEAD_CHUNK"); int v = e ? std::atoi(e) : 49152; return (v < 1024 || v > (1 << 24) || (v & 1023)) ? 49152 : v; }();  // a multiple of every batch size
    ph::RaySortParams sortp{};
    ph::RaySortGrid sort_grid{};
    if (sort_mode) {
        if ((rc = ensure_buf(s, w.d_sort_bins, 2 * PH_SORT_KEYS * 4))) return rc;
        sortp.order = (uint32_t*)w.d_order.p; sortp.bin_start = (uint32_t*)w.d_sort_bins.p; sortp.bin_cursor = sortp.bin_start + PH_SORT_KEYS;
        sortp.keys_cl = (const uint32_t*)w.d_keys_cl.p; sortp.keys_sh = (const uint32_t*)w.d_keys_sh.p;
        sort_grid.mode = (uint32_t)sort_mode;
        for (int k = 0; k < 3; k++) {
            const float ext = s->bvh.root_hi[k] - s->bvh.root_lo[k];
            sort_grid.lo[k] = s->bvh.root_lo[k]; sort_grid.scale[k] = ext > 0.0f ? 1.0f / ext : 0.0f;
        }
    }
    if (n_heads > 1 && (rc = ensure_buf(s, w.d_heads, (size_t)(n_iter_cap + 2) * 8 * 64))) return rc;
    ph::MatSortParams msp{};
    if (mat_queues) {
        if ((rc = ensure_buf(s, w.d_mbins, (size_t)(2 * PH_MS_BINS + 1) * 4))) return rc;
        msp.max_depth = max_depth; msp.mat_key = s->ds.mat_key; msp.key_emit = s->ds.ms_key_emit; msp.key_idle = s->ds.ms_key_idle;
        msp.tris = s->ds.tris; msp.meshes = s->ds.meshes; msp.hits = (const ph::HitOut*)w.d_hits.p;
        msp.keys = (uint16_t*)w.d_mkeys.p; msp.order = (uint32_t*)w.d_morder.p;
        msp.bin_start = (uint32_t*)w.d_mbins.p; msp.bin_cursor = msp.bin_start + PH_MS_BINS + 1;
    }

    ph::WfParams wp{};
    wp.cam = s->cam; wp.sp = s->sampler;
    wp.textured = s->textured_materials ? 1u : 0u; wp.cam_dev = nullptr;
    wp.any_rt = 0u;
    for (const MaterialRec& mr : s->materials) if (mr.rt_mode) wp.any_rt = 1u;
    if (s->textured_materials) {
        if ((rc = ensure_buf(s, w.d_cam, sizeof(CameraRec)))) return rc;
        PH_CHECK(s, hipMemcpyAsync(w.d_cam.p, &s->cam, sizeof(CameraRec), hipMemcpyHostToDevice, s->stream));
        wp.cam_dev = (const CameraRec*)w.d_cam.p;
    }
    for (int i = 0; i < 4; i++) wp.pixel_bounds[i] = pixel_bounds[i];
    wp.max_depth = max_depth; wp.rr_threshold = rr_threshold;
    wp.n_px = n_px; wp.px_xy = (const int2*)w.d_px.p;
    wp.rays_cl[0] = (ph::RayIn*)w.d_rays_cl[0].p; wp.rays_cl[1] = (ph::RayIn*)w.d_rays_cl[1].p;
    wp.hits_cl = (ph::HitOut*)w.d_hits.p; wp.rays_sh = (ph::RayIn*)w.d_rays_sh.p; wp.occ = (uint8_t*)w.d_occ.p;
    wp.live[0] = (uint32_t*)w.d_live[0].p; wp.live[1] = (uint32_t*)w.d_live[1].p;
    wp.ctr = (ph::IterCounters*)w.d_ctr.p; wp.stats = (ph::DevStats*)w.d_stats.p;
    for (int k = 0; k < 2; k++) {
        wp.s_L[k] = (float4*)w.d_sL.p + (size_t)k * B; wp.s_beta[k] = (float4*)w.d_sbeta.p + (size_t)k * B; wp.s_A[k] = (float4*)w.d_sA.p + (size_t)k * B;
        wp.s_f2[k] = (float4*)w.d_sf2.p + (size_t)k * B; wp.s_bold[k] = (float4*)w.d_sbold.p + (size_t)k * B; wp.s_idx[k] = (uint4*)w.d_sidx.p + (size_t)k * B;
        wp.s_prev[k] = (uint32_t*)w.d_sprev.p + (size_t)k * B;
    }
    wp.rec_L = (float4*)w.d_recL.p; wp.rec_py = (float*)w.d_recpy.p; wp.px_rounded = (uint8_t*)w.d_rounded.p;
    wp.sort_grid = sort_grid; wp.keys_cl = (uint32_t*)w.d_keys_cl.p; wp.keys_sh = (uint32_t*)w.d_keys_sh.p;
    wp.m_order = mat_queues ? (const uint32_t*)w.d_morder.p : nullptr; wp.m_bins = mat_queues ? (const uint32_t*)w.d_mbins.p : nullptr;

    if (spatial) { if ((rc = setup_spatial(s, wp.spatial))) return rc; }
    PH_CHECK(s, hipMemsetAsync(w.d_stats.p, 0, 64 + 6 * PHC_N * 8, s->stream));
    wp.phase = PH_PHASE_CLOCK ? (unsigned long long*)((char*)w.d_stats.p + 64) : nullptr;
    bool identity = true;  // does pixel_bounds cover every pixel of this rank's tiles?
    for (const ph::TileInfo& t : w.tiles)
        if (t.tb[0] < pixel_bounds[0] || t.tb[1] < pixel_bounds[1] || t.tb[2] > pixel_bounds[2] || t.tb[3] > pixel_bounds[3]) { identity = false; break; }
    size_t ev = 0;
    hipEvent_t e_begin = get_event(s, ev++), e_end = get_event(s, ev++);
    if (!e_begin || !e_end) return set_err(s, PBRT_HIP_ERR_DEVICE, "hipEventCreate failed");
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    auto timed = [&](int kind, auto&& launch) -> int {
        hipEvent_t a = get_event(s, ev++), b = get_event(s, ev++);
        if (!a || !b) return set_err(s, PBRT_HIP_ERR_DEVICE, "hipEventCreate failed");
        PH_CHECK(s, hipEventRecord(a, s->stream));
        launch();
        PH_CHECK(s, hipGetLastError());
        PH_CHECK(s, hipEventRecord(b, s->stream));
        spans.push_back({a, b, kind});
        return PBRT_HIP_OK;
    };
    PH_CHECK(s, hipEventRecord(e_begin, s->stream));
    static const bool split_traversal = std::getenv("PBRT_HIP_SPLIT_TRAVERSAL") != nullptr;  // measurement aid: one launch per ray kind
    uint64_t regular = 0, shadow = 0;
    std::vector<ph::IterCounters> hctr((size_t)n_iter_cap + 2);
    const uint32_t shade_blocks = (uint32_t)std::min<size_t>((B + 255) / 256, 256 * 16);
    wp.tex_out = nullptr;
    if (s->textured_materials) wp.tex_out = (TexOut*)w.d_tex_out.p;

    for (uint32_t s0 = 0; s0 < spp; s0 += chunk_spp) {
        const uint32_t cs = std::min(chunk_spp, spp - s0);
        wp.chunk_spp = cs; wp.s0 = s0; wp.B = n_px * cs; wp.identity_slots = identity ? 1u : 0u;
        PH_CHECK(s, hipMemsetAsync(w.d_ctr.p, 0, (size_t)(n_iter_cap + 2) * sizeof(ph::IterCounters), s->stream));
        if (n_heads > 1) PH_CHECK(s, hipMemsetAsync(w.d_heads.p, 0, (size_t)(n_iter_cap + 2) * 8 * 64, s->stream));
        if (identity) hipLaunchKernelGGL(ph::preset_counters_kernel, dim3(1), dim3(1), 0, s->stream, wp.ctr, wp.stats, wp.B);
        if ((rc = timed(2, [&]() { hipLaunchKernelGGL(ph::raygen_kernel, dim3((wp.B + 255) / 256), dim3(256), 0, s->stream, s->ds, wp); }))) return rc;
        int iters_run = 0;
        for (int it = 0; it < n_iter_cap; it++) {
            ph::IterCounters* c = (ph::IterCounters*)w.d_ctr.p + it;
            if (it >= n_iter) {  // only with "none" materials: go on while some path is still alive
                uint32_t live = 0;
                PH_CHECK(s, hipMemcpyAsync(&live, &c->n_live, 4, hipMemcpyDeviceToHost, s->stream));
                PH_CHECK(s, hipStreamSynchronize(s->stream));
                if (live == 0) break;
                if (it == n_iter_cap - 1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "render: a path crossed more than 1024 'none' surfaces");
            }
            iters_run = it + 1;
            ph::TravParams tp{};
            tp.rays = wp.rays_cl[it & 1]; tp.out = wp.hits_cl; tp.n = 0; tp.n_ptr = &c->n_cl; tp.counter = &c->head_cl;
            if (n_heads > 1) { tp.heads = (uint32_t*)w.d_heads.p + (size_t)it * 8 * 16; tp.n_heads = (uint32_t)n_heads; tp.head_chunk = (uint32_t)head_chunk; }
            if (it > 0 && !split_traversal) {  // this round's extension rays and the shadow rays of the previous vertices: one launch, one tail
                tp.rays2 = wp.rays_sh; tp.out2 = wp.occ; tp.n2_ptr = &c->n_sh;
                if (sort_mode) {  // regroup the round's rays by origin cell; camera rays (round 0) leave raygen in pixel order already
                    sortp.n_cl = &c->n_cl; sortp.n_sh = &c->n_sh;
                    if ((rc = timed(3, [&]() {
                            (void)hipMemsetAsync(sortp.bin_start, 0, PH_SORT_KEYS * 4, s->stream);
                            hipLaunchKernelGGL(ph::raysort_hist_kernel, dim3(sort_blocks), dim3(PH_SORT_BLOCK), 0, s->stream, sortp);
                            hipLaunchKernelGGL(ph::raysort_scan_kernel, dim3(1), dim3(1024), 0, s->stream, sortp);
                            hipLaunchKernelGGL(ph::raysort_scatter_kernel, dim3(sort_blocks), dim3(PH_SORT_BLOCK), 0, s->stream, sortp);
                        }))) return rc;
                    tp.order = sortp.order;
                }
                if ((rc = timed(0, [&]() { launch_traverse_kernel(s, 2, s->trav_blocks, tp); }))) return rc;
            } else {
                if ((rc = timed(0, [&]() { launch_traverse_kernel(s, 0, s->trav_blocks, tp); }))) return rc;
                if (it > 0) {
                    tp.rays = wp.rays_sh; tp.out = wp.occ; tp.n_ptr = &c->n_sh; tp.counter = &c->head_sh;
                    tp.heads = nullptr; tp.n_heads = 0;   // the round's queue heads were drained by the closest-hit launch: this one pulls from its own single head
                    if ((rc = timed(1, [&]() { launch_traverse_kernel(s, 1, s->trav_blocks, tp); }))) return rc;
                }
            }
            // the shade side's work queues: this round's list regrouped by what has to be done for each path (matsort.h)
            if (mat_queues) {
                msp.s_idx = wp.s_idx[it & 1]; msp.n_live = &c->n_live;
                if ((rc = timed(2, [&]() {
                        (void)hipMemsetAsync(msp.bin_start, 0, (PH_MS_BINS + 1) * 4, s->stream);
                        hipLaunchKernelGGL(ph::matsort_hist_kernel, dim3(sort_blocks), dim3(PH_MS_BLOCK), 0, s->stream, msp);
                        hipLaunchKernelGGL(ph::matsort_scan_kernel, dim3(1), dim3(PH_MS_BINS), 0, s->stream, msp);
                        hipLaunchKernelGGL(ph::matsort_scatter_kernel, dim3(sort_blocks), dim3(PH_MS_BLOCK), 0, s->stream, msp);
                    }))) return rc;
            }
            // the texture pass first: besides colours and bumped frames it finds the hits that have no BSDF at all (TexOut::bumped, PH_TEXOUT_NULL_BSDF), which the
            // light-distribution pass must skip as the reference's `continue` does (path.rs:142-157)
            if (s->textured_materials) {
                if ((rc = timed(2, [&]() {
                        // waves per SIMD the variants are compiled for (PBRT_HIP_TEX_WAVES = "<camera round><later rounds>", e.g. 23; A/B aid)
                        static const int tw = []() { const char* e = std::getenv("PBRT_HIP_TEX_WAVES"); const int v = e ? std::atoi(e) : 0; return (v / 10 >= 2 && v / 10 <= 3 && v % 10 >= 2 && v % 10 <= 4) ? v : 0; }();
                        const dim3 g(shade_blocks), b(256);
                        if (it == 0) {   // camera rays: differentials, filtered look-ups
                            if (s->simple_textures) hipLaunchKernelGGL((ph::texture_kernel<true, true, 3>), g, b, 0, s->stream, s->ds, wp, it);
                            else if (tw / 10 == 3) hipLaunchKernelGGL((ph::texture_kernel<false, true, 3>), g, b, 0, s->stream, s->ds, wp, it);
                            else hipLaunchKernelGGL((ph::texture_kernel<false, true, 2>), g, b, 0, s->stream, s->ds, wp, it);
                        } else {
                            if (s->simple_textures) { if (tw % 10 == 3) hipLaunchKernelGGL((ph::texture_kernel<true, false, 3>), g, b, 0, s->stream, s->ds, wp, it); else hipLaunchKernelGGL((ph::texture_kernel<true, false, 4>), g, b, 0, s->stream, s->ds, wp, it); }
                            else if (tw % 10 == 2) hipLaunchKernelGGL((ph::texture_kernel<false, false, 2>), g, b, 0, s->stream, s->ds, wp, it);
                            else if (tw % 10 == 4) hipLaunchKernelGGL((ph::texture_kernel<false, false, 4>), g, b, 0, s->stream, s->ds, wp, it);
                            else hipLaunchKernelGGL((ph::texture_kernel<false, false, 3>), g, b, 0, s->stream, s->ds, wp, it);
                        }
                    }))) return rc;
            }
            if (spatial && (it < max_depth || s->has_none_material)) {  // vertices reached at bounce == max_depth sample no light (path.rs:136-139)
                if ((rc = timed(2, [&]() {
                        hipLaunchKernelGGL(ph::spatial_mark_kernel, dim3(shade_blocks), dim3(256), 0, s->stream, s->ds, wp, it);
                        hipLaunchKernelGGL(ph::spatial_compute_kernel, dim3(1024), dim3(PH_SPATIAL_BLOCK), 0, s->stream, s->ds, wp.spatial);
                    }))) return rc;
            }
            if ((rc = timed(2, [&]() {
                    if (s->textured_materials) {
                        if (s->general_materials) hipLaunchKernelGGL((ph::shade_kernel<true, true>), dim3(shade_blocks), dim3(256), 0, s->stream, s->ds, wp, it);
                        else hipLaunchKernelGGL((ph::shade_kernel<false, true>), dim3(shade_blocks), dim3(256), 0, s->stream, s->ds, wp, it);
                    } else if (s->general_materials) hipLaunchKernelGGL(ph::shade_kernel<true>, dim3(shade_blocks), dim3(256), 0, s->stream, s->ds, wp, it);
                    else hipLaunchKernelGGL(ph::shade_kernel<false>, dim3(shade_blocks), dim3(256), 0, s->stream, s->ds, wp, it);
                }))) return rc;
        }
        PH_CHECK(s, hipMemcpyAsync(hctr.data(), w.d_ctr.p, (size_t)(iters_run + 1) * sizeof(ph::IterCounters), hipMemcpyDeviceToHost, s->stream));
        PH_CHECK(s, hipStreamSynchronize(s->stream));
        for (int it = 0; it < iters_run; it++) { regular += hctr[it].n_cl; shadow += hctr[it].n_sh; }
    }

    // ---- film: per-tile accumulation in reference order ---------------------------------------------------------------------------
    ph::FilmParams fp{};
    fp.film = s->film; fp.tiles = (const ph::TileInfo*)w.d_tiles.p; fp.n_tiles = (uint32_t)w.tiles.size();
    fp.slot_w = w.slot_w; fp.slot_h = w.slot_h; fp.spp = spp; fp.n_px = n_px; fp.rec_L = wp.rec_L; fp.rec_py = wp.rec_py; fp.px_rounded = wp.px_rounded; fp.tile_buf = (float4*)d_tile_buffer;
    const uint64_t film_threads = (uint64_t)fp.n_tiles * w.slot_w * w.slot_h;
    if ((rc = timed(2, [&]() { hipLaunchKernelGGL(ph::film_tiles_kernel, dim3((uint32_t)((film_threads + 255) / 256)), dim3(256), 0, s->stream, fp); }))) return rc;
    PH_CHECK(s, hipEventRecord(e_end, s->stream));
    PH_CHECK(s, hipStreamSynchronize(s->stream));

    uint32_t flag = 0;
    PH_CHECK(s, hipMemcpy(&flag, s->d_error.p, 4, hipMemcpyDeviceToHost));
    if (flag) { (void)hipMemset(s->d_error.p, 0, 4); return set_err(s, PBRT_HIP_ERR_DEVICE, "traversal stack exceeded 64 entries (the reference panics here, bvh/mod.rs:185)"); }
    uint32_t sp_ctr[4] = {0, 0, 0, 0};
    if (spatial) {
        PH_CHECK(s, hipMemcpy(sp_ctr, w.d_sp_ctr.p, sizeof sp_ctr, hipMemcpyDeviceToHost));
        if (sp_ctr[ph::SP_OVERFLOW])
            return set_err(s, PBRT_HIP_ERR_OOM, "render: SpatialLightDistribution pool exhausted (" + std::to_string(sp_ctr[ph::SP_CLAIMED]) + " voxels x " +
                                                std::to_string(s->lights.size()) + " lights); use lightsamplestrategy power/uniform or a larger GPU memory budget");
    }
#if PH_PHASE_CLOCK
    {   // measurement build: the shade kernel's phase clocks of this render, to stderr
        unsigned long long ph_[3 * PHC_N];
        PH_CHECK(s, hipMemcpy(ph_, (const char*)w.d_stats.p + 64, sizeof ph_, hipMemcpyDeviceToHost));
        static const char* names[11] = {"state+resolve", "surface+emission", "bsdf+textures", "sampler_dims", "light_pick", "light_sample_li", "f/pdf+shadow_ray", "mis_half", "bsdf_sample+rr", "queue_append", "whole_kernel"};
        for (int k = 0; k < 11; k++)
            std::fprintf(stderr, "SHADE_PHASE %-17s cycles %14llu (%5.1f %% of the kernel)  executions %12llu  mean active lanes %5.1f\n", names[k], ph_[k], ph_[10] ? 100.0 * (double)ph_[k] / (double)ph_[10] : 0.0,
                         ph_[PHC_N + k], ph_[PHC_N + k] ? (double)ph_[2 * PHC_N + k] / (double)ph_[PHC_N + k] : 0.0);
        unsigned long long pt_[3 * PHC_N];
        PH_CHECK(s, hipMemcpy(pt_, (const char*)w.d_stats.p + 64 + 3 * PHC_N * 8, sizeof pt_, hipMemcpyDeviceToHost));
        static const char* tnames[4] = {"hit+context", "bump", "lobe_colours", "whole_kernel"};
        for (int k = 0; k < 4; k++)
            std::fprintf(stderr, "TEXTURE_PHASE %-13s cycles %14llu (%5.1f %% of the kernel)  executions %12llu  mean active lanes %5.1f\n", tnames[k], pt_[k], pt_[3] ? 100.0 * (double)pt_[k] / (double)pt_[3] : 0.0,
                         pt_[PHC_N + k], pt_[PHC_N + k] ? (double)pt_[2 * PHC_N + k] / (double)pt_[PHC_N + k] : 0.0);
    }
#endif
    if (out_stats) {
        out_stats->light_distributions_created = sp_ctr[ph::SP_DONE];
        ph::DevStats ds;
        PH_CHECK(s, hipMemcpy(&ds, w.d_stats.p, sizeof(ds), hipMemcpyDeviceToHost));
        out_stats->camera_rays = ds.camera_rays; out_stats->regular_rays = regular; out_stats->shadow_rays = shadow;
        out_stats->paths_total = ds.paths_total; out_stats->paths_zero_radiance = ds.paths_zero;
        float ms = 0;
        PH_CHECK(s, hipEventElapsedTime(&ms, e_begin, e_end));
        out_stats->render_seconds = ms * 1e-3;
        double acc[4] = {0, 0, 0, 0};
        for (const Span& sp : spans) { float m = 0; if (hipEventElapsedTime(&m, sp.a, sp.b) == hipSuccess) acc[sp.kind] += m * 1e-3; }
        out_stats->extend_seconds = acc[0]; out_stats->shadow_seconds = acc[1]; out_stats->shade_seconds = acc[2] + acc[3];  // ray binning counts as non-traversal time
        for (const Span& sp : spans) { if (sp.kind == 0) out_stats->extend_launches++; else if (sp.kind == 1) out_stats->shadow_launches++; }
    }
    return PBRT_HIP_OK;
}

int merge_tiles(PbrtHipScene* s, int tile_size, int parts, const void* const* d_bufs, float* out_xyz, float* out_weight) {
    if (!s->wf) s->wf = new Wavefront();
    Wavefront& w = *s->wf;
    const FilmRec& f = s->film;
    const TileGrid g = tile_grid(f, tile_size);  // the merge needs the frame-wide grid only, not a rank's tile list
    const int cw = f.crop[2] - f.crop[0], ch = f.crop[3] - f.crop[1];
    const size_t npx = (size_t)std::max(cw, 0) * (size_t)std::max(ch, 0);
    if (npx == 0) return PBRT_HIP_OK;
    int rc;
    if ((rc = ensure_buf(s, w.d_xyz, npx * 12))) return rc;
    if ((rc = ensure_buf(s, w.d_w, npx * 4))) return rc;
    ph::MergeParams mp{};
    mp.film = f;
    for (int i = 0; i < 4; i++) mp.sb[i] = g.sb[i];
    mp.tile_size = tile_size; mp.ntx = g.ntx; mp.nty = g.nty; mp.parts = parts; mp.slot_w = g.slot_w; mp.slot_h = g.slot_h;
    for (int i = 0; i < parts; i++) mp.bufs[i] = (const float4*)d_bufs[i];
    mp.out_xyz = (float*)w.d_xyz.p; mp.out_w = (float*)w.d_w.p;
    hipLaunchKernelGGL(ph::merge_kernel, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, s->stream, mp);
    PH_CHECK(s, hipGetLastError());
    PH_CHECK(s, hipMemcpyAsync(out_xyz, w.d_xyz.p, npx * 12, hipMemcpyDeviceToHost, s->stream));
    PH_CHECK(s, hipMemcpyAsync(out_weight, w.d_w.p, npx * 4, hipMemcpyDeviceToHost, s->stream));
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    return PBRT_HIP_OK;
}

int spatial_voxels_touched(PbrtHipScene* s, std::vector<uint8_t>& touched, uint64_t* count) {
    if (!s->wf || !s->wf->d_vox_slot.p || !s->wf->n_vox) return PBRT_HIP_OK;
    PH_CHECK(s, hipSetDevice(s->device));
    std::vector<int32_t> slot(s->wf->n_vox);
    PH_CHECK(s, hipMemcpy(slot.data(), s->wf->d_vox_slot.p, slot.size() * 4, hipMemcpyDeviceToHost));
    if (touched.size() < slot.size()) touched.resize(slot.size(), 0);
    for (size_t v = 0; v < slot.size(); v++)
        if (slot[v] >= 0 && !touched[v]) { touched[v] = 1; (*count)++; }
    return PBRT_HIP_OK;
}

DevBuf& tile_buffer_of(PbrtHipScene* s) {
    if (!s->wf) s->wf = new Wavefront();
    return s->wf->d_tilebuf;
}

}  // namespace phost

using namespace phost;

extern "C" {

int pbrt_hip_tile_buffer_floats(PbrtHipScene* s, int tile_size, int tile_part, int tile_parts, uint64_t* out_floats) {
    return ph_guard(s, "pbrt_hip_tile_buffer_floats", [&]() -> int {
    if (!s || !out_floats) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->have_film) return set_err(s, PBRT_HIP_ERR_STATE, "tile_buffer_floats: set_film first");
    if (tile_size <= 0 || tile_parts <= 0 || tile_part < 0 || tile_part >= tile_parts) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "tile_buffer_floats: bad partition");
    *out_floats = tile_buffer_floats_for(s, tile_size, tile_part, tile_parts);
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_render_path_tiles_device(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size,
                                      int tile_part, int tile_parts, void* d_tile_buffer, PbrtHipStats* out_stats) {
    return ph_guard(s, "pbrt_hip_render_path_tiles_device", [&]() -> int {
    int rc = check_render_args(s, max_depth, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts);
    if (rc) return rc;
    if (!d_tile_buffer) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: null tile buffer");
    return render_tiles(s, max_depth, rr_threshold, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts, d_tile_buffer, out_stats);
    });
}

int pbrt_hip_merge_tiles_device(PbrtHipScene* s, int tile_size, int tile_parts, const void* const* d_tile_buffers, float* out_xyz, float* out_weight) {
    return ph_guard(s, "pbrt_hip_merge_tiles_device", [&]() -> int {
    if (!s || !d_tile_buffers || !out_xyz || !out_weight) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "merge: null argument");
    if (!s->have_film) return set_err(s, PBRT_HIP_ERR_STATE, "merge: set_film first");
    if (tile_size <= 0 || tile_parts <= 0 || tile_parts > PH_MAX_TILE_PARTS) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "merge: bad partition");
    PH_CHECK(s, hipSetDevice(s->device));
    return merge_tiles(s, tile_size, tile_parts, d_tile_buffers, out_xyz, out_weight);
    });
}

int pbrt_hip_render_path(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size, int tile_part,
                         int tile_parts, float* out_xyz, float* out_weight, PbrtHipStats* out_stats) {
    return ph_guard(s, "pbrt_hip_render_path", [&]() -> int {
    int rc = check_render_args(s, max_depth, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts);
    if (rc) return rc;
    if (!out_xyz || !out_weight) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: null output");
    if (s->multi) return render_path_multi(s, max_depth, rr_threshold, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts, out_xyz, out_weight, out_stats);
    if (!s->wf) s->wf = new Wavefront();
    const size_t floats = tile_buffer_floats_for(s, tile_size, tile_part, tile_parts);
    if ((rc = ensure_buf(s, s->wf->d_tilebuf, floats * 4))) return rc;
    if ((rc = render_tiles(s, max_depth, rr_threshold, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts, s->wf->d_tilebuf.p, out_stats))) return rc;
    // a single rank holds only its own tiles: the other parts contribute nothing (zero buffers are not needed: merge
    // addresses only tiles t with t % parts == part when every other part's pointer aliases an all-zero slot)
    if (tile_parts == 1) {
        const void* bufs[1] = {s->wf->d_tilebuf.p};
        return merge_tiles(s, tile_size, 1, bufs, out_xyz, out_weight);
    }
    // partial frame: merge this part against zeroed stand-ins for the missing ones
    std::vector<DevBuf> zeros((size_t)tile_parts);
    std::vector<const void*> bufs((size_t)tile_parts);
    for (int i = 0; i < tile_parts; i++) {
        if (i == tile_part) { bufs[i] = s->wf->d_tilebuf.p; continue; }
        const size_t fl = tile_buffer_floats_for(s, tile_size, i, tile_parts);
        if (hipMalloc(&zeros[i].p, fl * 4) != hipSuccess) { for (auto& z : zeros) if (z.p) (void)hipFree(z.p); return set_err(s, PBRT_HIP_ERR_OOM, "render: out of device memory"); }
        (void)hipMemset(zeros[i].p, 0, fl * 4);
        bufs[i] = zeros[i].p;
    }
    // setup_tiles(part) was used for rendering; merge needs the frame-wide tile grid only (sb/ntx/nty/slots are part-independent)
    rc = merge_tiles(s, tile_size, tile_parts, bufs.data(), out_xyz, out_weight);
    for (auto& z : zeros) if (z.p) (void)hipFree(z.p);
    return rc;
    });
}

int pbrt_hip_generate_camera_rays(PbrtHipScene* s, const int pb[4], uint32_t sample_index, PbrtHipRay* out_rays, float* out_pfilm) {
    return ph_guard(s, "pbrt_hip_generate_camera_rays", [&]() -> int {
    if (!s || !pb || !out_rays) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "generate_camera_rays: null argument");
    if (!s->have_camera || !s->have_sampler) return set_err(s, PBRT_HIP_ERR_STATE, "generate_camera_rays: camera and sampler must be set");
    if (s->sampler.kind == 1 && s->sobol32.empty()) return set_err(s, PBRT_HIP_ERR_STATE, "generate_camera_rays: sobol tables not set");
    PH_CHECK(s, hipSetDevice(s->device));
    const int wdt = pb[2] - pb[0], hgt = pb[3] - pb[1];
    if (wdt <= 0 || hgt <= 0) return PBRT_HIP_OK;
    const size_t n = (size_t)wdt * hgt;
    int rc;
    // camera rays need only the sampler tables; upload_scene copes with a scene whose BVH has not been built
    if ((rc = upload_scene(s))) return rc;
    if ((rc = ensure_buf(s, s->d_rays_tmp, n * sizeof(PbrtHipRay)))) return rc;
    if ((rc = ensure_buf(s, s->d_out_tmp, n * 8))) return rc;
    hipLaunchKernelGGL(ph::camera_rays_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s->stream, s->ds, s->cam, s->sampler, pb[0], pb[1], pb[2], pb[3],
                       sample_index, (ph::RayIn*)s->d_rays_tmp.p, (float*)s->d_out_tmp.p);
    PH_CHECK(s, hipGetLastError());
    PH_CHECK(s, hipMemcpyAsync(out_rays, s->d_rays_tmp.p, n * sizeof(PbrtHipRay), hipMemcpyDeviceToHost, s->stream));
    if (out_pfilm) PH_CHECK(s, hipMemcpyAsync(out_pfilm, s->d_out_tmp.p, n * 8, hipMemcpyDeviceToHost, s->stream));
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    return PBRT_HIP_OK;
    });
}

}  // extern "C"
