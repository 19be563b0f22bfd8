// Device-side restatement of the per-vertex work of PathIntegrator::li: samplers, camera, surface set-up, matte BSDF,
// lights.  Every function cites the reference code it follows; expression order is the reference's (f32, no FMA).
#pragma once
#include "dmath.h"
#include "scene_types.h"
#include "traverse.h"

namespace ph {

struct f2 { float x, y; };
PH_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }

// =============================== samplers =========================================================================
// radical_inverse base 2 (core/src/low_discrepency.rs:456-460): reverse_bits_64(a) as f32 * 2^-64, no clamp
PH_DEV float radical_inverse_2(uint64_t a) {
    uint64_t r = ((uint64_t)__brev((uint32_t)a) << 32) | (uint64_t)__brev((uint32_t)(a >> 32));
    return __ull2float_rn(r) * 0x1.0p-64f;
}
// radical_inverse_specialized (:401-421)
PH_DEV float radical_inverse_base(uint32_t base, uint32_t a) {
    const float inv_base = ph_div(1.0f, (float)base);
    uint64_t reversed = 0;
    float inv_base_n = 1.0f;
    while (a != 0) {
        uint32_t next = a / base, digit = a - next * base;
        reversed = reversed * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return pminf(__ull2float_rn(reversed) * inv_base_n, kOneMinusEps);
}
// scrambled_radical_inverse_specialized (:428-449).  `a / base` is the integer quotient; it is evaluated as
// umul64hi(a, ceil(2^64/base)), which equals floor(a/base) for every a < 2^32 (integer arithmetic: no rounding involved).
PH_DEV float scrambled_radical_inverse(uint32_t base, uint64_t magic, uint32_t a, const uint16_t* __restrict__ perm) {
    const float inv_base = ph_div(1.0f, (float)base);
    uint64_t reversed = 0;
    float inv_base_n = 1.0f;
    while (a != 0) {
        uint32_t next = (uint32_t)__umul64hi((uint64_t)a, magic), digit = a - next * base;
        reversed = reversed * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return pminf(inv_base_n * (__ull2float_rn(reversed) + ph_div(inv_base * (float)perm[0], 1.0f - inv_base)), kOneMinusEps);
}
// inverse_radical_inverse (:1535-1545)
PH_DEV uint32_t inverse_radical_inverse(uint32_t base, uint32_t inverse, uint32_t n_digits) {
    uint32_t index = 0;
    for (uint32_t i = 0; i < n_digits; i++) { uint32_t digit = inverse % base; inverse /= base; index = index * base + digit; }
    return index;
}
PH_DEV int rem_i(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }  // core/src/pbrt/common.rs:116-126

// HaltonSampler::get_index_for_sample (samplers/src/halton.rs:118-144): the per-pixel offset via the CRT
PH_DEV uint32_t halton_pixel_offset(const SamplerRec& sp, int px, int py) {
    uint64_t off = 0;
    if (sp.sample_stride > 1) {
        const int pm[2] = {rem_i(px, 128), rem_i(py, 128)};
        for (int i = 0; i < 2; i++) {
            uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2u : 3u, (uint32_t)pm[i], sp.base_exponents[i]);
            off += dim_offset * (uint64_t)(sp.sample_stride / sp.base_scales[i]) * (uint64_t)sp.mult_inverse[i];
        }
        off %= sp.sample_stride;
    }
    return (uint32_t)off;
}
// The first PH_LDS_DIMS dimensions' tables (6.7 KB of digit permutations + primes + magic numbers) can be staged in LDS:
// a path vertex draws ~8 dimensions x ~7 digits, i.e. ~80 dependent table reads.  `lds` is null when a kernel does not stage them.
// 54 dimensions = every prime below 256, so a permutation entry fits a byte (maxdepth 5 uses dimensions 0..52; deeper paths read
// the later dimensions from the global tables).  LDS per block decides how many shade blocks a CU holds.
#define PH_LDS_DIMS 54
#define PH_LDS_PERMS 6081  // sum of the first 54 primes
struct HaltonLds {
    uint8_t perms[PH_LDS_PERMS + 2];
    uint32_t primes[PH_LDS_DIMS], sums[PH_LDS_DIMS];
    uint64_t magic[PH_LDS_DIMS];
};
PH_DEV void halton_lds_fill(HaltonLds* l, const DeviceScene& sc) {  // cooperative; caller synchronises
    for (uint32_t i = threadIdx.x; i < PH_LDS_PERMS; i += blockDim.x) l->perms[i] = (uint8_t)sc.halton_perms[i];
    for (uint32_t i = threadIdx.x; i < PH_LDS_DIMS; i += blockDim.x) { l->primes[i] = sc.primes[i]; l->sums[i] = sc.prime_sums[i]; l->magic[i] = sc.prime_magic[i]; }
}
// scrambled radical inverse reading the permutation from LDS
PH_DEV float scrambled_radical_inverse_lds(const HaltonLds* l, uint32_t dim, uint32_t a) {
    const uint32_t base = l->primes[dim];
    const uint64_t magic = l->magic[dim];
    const uint8_t* perm = l->perms + l->sums[dim];
    const float inv_base = ph_div(1.0f, (float)base);
    uint64_t reversed = 0;
    float inv_base_n = 1.0f;
    while (a != 0) {
        uint32_t next = (uint32_t)__umul64hi((uint64_t)a, magic), digit = a - next * base;
        reversed = reversed * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return pminf(inv_base_n * (__ull2float_rn(reversed) + ph_div(inv_base * (float)perm[0], 1.0f - inv_base)), kOneMinusEps);
}
// HaltonSampler::sample_dimension (halton.rs:146-160)
PH_DEV float halton_sample(const DeviceScene& sc, const SamplerRec& sp, uint32_t index, uint32_t dim, const HaltonLds* lds = nullptr) {
    if (sp.at_center && (dim == 0 || dim == 1)) return 0.5f;
    if (dim == 0) return radical_inverse_2((uint64_t)(index >> sp.base_exponents[0]));
    if (dim == 1) return radical_inverse_base(3u, index / sp.base_scales[1]);
    if (lds && dim < PH_LDS_DIMS) return scrambled_radical_inverse_lds(lds, dim, index);
    return scrambled_radical_inverse(sc.primes[dim], sc.prime_magic[dim], index, sc.halton_perms + sc.prime_sums[dim]);
}

// sobol_interval_to_index (core/src/low_discrepency.rs:1770-1810)
PH_DEV uint64_t sobol_interval_to_index(const DeviceScene& sc, uint32_t m, uint64_t frame, int px, int py) {
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (int c = 0; frame != 0; frame >>= 1, c++)
        if (frame & 1) delta ^= sc.vdc[(size_t)(m - 1) * 52 + c];
    uint64_t b = ((((uint64_t)(uint32_t)px) << m) | ((uint64_t)(uint32_t)py)) ^ delta;
    for (int c = 0; b != 0; b >>= 1, c++)
        if (b & 1) index ^= sc.vdc_inv[(size_t)(m - 1) * 52 + c];
    return index;
}
// sobol_sample_f32 with scramble 0 (:1826-1848) + SobolSampler::sample_dimension (samplers/src/sobol.rs:77-93)
PH_DEV float sobol_sample(const DeviceScene& sc, const SamplerRec& sp, uint64_t a, uint32_t dim, int px, int py) {
    uint32_t v = 0;
    for (size_t i = (size_t)dim * 52; a != 0; a >>= 1, i++)
        if (a & 1) v ^= sc.sobol32[i];
    float s = pminf((float)v * 0x1.0p-32f, kOneMinusEps);
    if (dim == 0 || dim == 1) {
        s = s * (float)sp.resolution + (float)sp.bounds[dim];
        s = pclampf(s - (float)(dim == 0 ? px : py), 0.0f, kOneMinusEps);
    }
    return s;
}

// A GlobalSampler cursor (core/src/sampler/common.rs:120-135): index of the current sample + next dimension.
struct SamplerCursor {
    uint64_t index;
    uint32_t dim;
    int px, py;
    const HaltonLds* lds;  // LDS-staged Halton tables, or null
};
PH_DEV float sampler_dim(const DeviceScene& sc, const SamplerRec& sp, const SamplerCursor& c, uint32_t dim) {
    return sp.kind == 0 ? halton_sample(sc, sp, (uint32_t)c.index, dim, c.lds) : sobol_sample(sc, sp, c.index, dim, c.px, c.py);
}
PH_DEV float get_1d(const DeviceScene& sc, const SamplerRec& sp, SamplerCursor& c) {  // halton.rs:226-235 (no sample arrays on this path)
    float p = sampler_dim(sc, sp, c, c.dim);
    c.dim += 1;
    return p;
}
PH_DEV f2 get_2d(const DeviceScene& sc, const SamplerRec& sp, SamplerCursor& c) {  // halton.rs:237-251
    if (c.dim + 1 >= 5 && c.dim < 5) c.dim = 5;
    f2 p = mk2(sampler_dim(sc, sp, c, c.dim), sampler_dim(sc, sp, c, c.dim + 1));
    c.dim += 2;
    return p;
}

// =============================== sampling routines (core/src/sampling/common.rs) ==================================
PH_DEV f2 concentric_sample_disk(f2 u) {  // :138-155
    f2 uo = mk2(2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f);
    if (uo.x == 0.0f && uo.y == 0.0f) return mk2(0.0f, 0.0f);
    float r, theta;
    if (pabs(uo.x) > pabs(uo.y)) { r = uo.x; theta = kPiOver4 * ph_div(uo.y, uo.x); }
    else { r = uo.y; theta = kPiOver2 - kPiOver4 * ph_div(uo.x, uo.y); }
    float sn, cs;
    d_sincos(theta, sn, cs);
    return mk2(r * cs, r * sn);
}
PH_DEV f3 cosine_sample_hemisphere(f2 u) {  // :207-211
    f2 d = concentric_sample_disk(u);
    float z = ph_sqrt(pmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return mk3(d.x, d.y, z);
}
PH_DEV f2 uniform_sample_triangle(f2 u) { float su0 = ph_sqrt(u.x); return mk2(1.0f - su0, u.y * su0); }  // :198-201
PH_DEV float power_heuristic1(float fp, float gp) {  // :239-243 with nf = ng = 1
    float f = 1.0f * fp, g = 1.0f * gp;
    return ph_div(f * f, f * f + g * g);
}
// find_interval (core/src/pbrt/common.rs:251-276) over cdf[i] <= u
PH_DEV uint32_t find_interval_cdf(const float* cdf, uint32_t size, float u) {
    uint32_t first = 0, len = size;
    while (len > 0) {
        uint32_t half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    if (first == 0) return 0;
    uint32_t v = first - 1, hi = size - 2;
    return v > hi ? hi : v;
}

// =============================== camera ============================================================================
// PerspectiveCamera / OrthographicCamera::generate_ray_differential's main ray (cameras/src/perspective_camera.rs:144-171, orthographic_camera.rs:121-149) followed by
// Transform::transform_ray (core/src/geometry/transform.rs:451-476, incl. quirk B2 t_max -= dt).  Differentials are
// consumed only by textures: the texture pass rebuilds them at a camera ray's first hit (texture.h: camera_ray_differentials).
// EnvironmentCamera's direction (environment_camera.rs:61-66): the whole sphere, y up in camera space.  Out of line: four f64 trigonometric
// evaluations that the projective cameras' ray generators should not carry inline
__device__ __noinline__ void environment_camera_dir(float px, float py, float xres, float yres, f3* out) {
    const float theta = ph_div(kPi * py, yres), phi = ph_div(kTwoPi * px, xres);
    const float st = d_sin(theta);
    *out = mk3(st * d_cos(phi), d_cos(theta), st * d_sin(phi));
}
PH_DEV void generate_camera_ray(const CameraRec& cam, f2 p_film, float time_s, f2 lens_s, RayIn& out) {
    const float* m = cam.r2c;
    float xp = m[0] * p_film.x + m[1] * p_film.y + m[2] * 0.0f + m[3];
    float yp = m[4] * p_film.x + m[5] * p_film.y + m[6] * 0.0f + m[7];
    float zp = m[8] * p_film.x + m[9] * p_film.y + m[10] * 0.0f + m[11];
    float wp = m[12] * p_film.x + m[13] * p_film.y + m[14] * 0.0f + m[15];
    f3 p_camera = (wp == 1.0f) ? mk3(xp, yp, zp) : mk3(xp, yp, zp) / wp;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = normalize(p_camera);
    if (cam.kind == PH_CAM_ORTHOGRAPHIC) { o = p_camera; d = mk3(0.0f, 0.0f, 1.0f); }  // orthographic_camera.rs:127-135: parallel rays from the film point
    if (cam.kind == PH_CAM_ENVIRONMENT) environment_camera_dir(p_film.x, p_film.y, cam.full_res[0], cam.full_res[1], &d);   // no lens
    const float time = (1.0f - time_s) * cam.shutter_open + time_s * cam.shutter_close;  // lerp (pbrt/common.rs:166-175)
    if (cam.lens_radius > 0.0f) {
        f2 cd = concentric_sample_disk(lens_s);
        f2 p_lens = mk2(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
        float ft = ph_div(cam.focal_distance, d.z);
        f3 p_focus = o + d * ft;
        o = mk3(p_lens.x, p_lens.y, 0.0f);
        d = normalize(p_focus - o);
    }
    const float* c = cam.c2w;
    // transform_point_with_error (transform.rs:304-328)
    float ox = (c[0] * o.x + c[1] * o.y) + (c[2] * o.z + c[3]);
    float oy = (c[4] * o.x + c[5] * o.y) + (c[6] * o.z + c[7]);
    float oz = (c[8] * o.x + c[9] * o.y) + (c[10] * o.z + c[11]);
    float ow = (c[12] * o.x + c[13] * o.y) + (c[14] * o.z + c[15]);
    float xs = pabs(c[0] * o.x) + pabs(c[1] * o.y) + pabs(c[2] * o.z) + pabs(c[3]);
    float ys = pabs(c[4] * o.x) + pabs(c[5] * o.y) + pabs(c[6] * o.z) + pabs(c[7]);
    float zs = pabs(c[8] * o.x) + pabs(c[9] * o.y) + pabs(c[10] * o.z) + pabs(c[11]);
    f3 o_err = kGamma3 * mk3(xs, ys, zs);
    f3 ow3 = (ow == 1.0f) ? mk3(ox, oy, oz) : mk3(ox, oy, oz) / ow;
    f3 dw = mk3(c[0] * d.x + c[1] * d.y + c[2] * d.z, c[4] * d.x + c[5] * d.y + c[6] * d.z, c[8] * d.x + c[9] * d.y + c[10] * d.z);
    float l2 = length_squared(dw), t_max = kInf;
    if (l2 > 0.0f) {
        float dt = ph_div(dot(vabs(dw), o_err), l2);
        ow3 = ow3 + dw * dt;
        t_max -= dt;
    }
    out.ox = ow3.x; out.oy = ow3.y; out.oz = ow3.z; out.t_max = t_max; out.dx = dw.x; out.dy = dw.y; out.dz = dw.z; out.time = time;
}

// =============================== surface interaction ===============================================================
struct SurfHit {
    f3 p, p_error, wo, n;  // Hit (core/src/interaction/mod.rs:107-125)
    f3 ns, dpdu_s;         // shading.n, shading.dpdu
    float time;
    uint32_t prim;
};
struct TriVerts { f3 p0, p1, p2; uint32_t i0, i1, i2; };
PH_DEV TriVerts load_tri(const DeviceScene& sc, uint32_t prim) {
    TriVerts t;
    t.i0 = sc.idx[3 * prim]; t.i1 = sc.idx[3 * prim + 1]; t.i2 = sc.idx[3 * prim + 2];
    t.p0 = ld3(sc.P + 3 * (size_t)t.i0); t.p1 = ld3(sc.P + 3 * (size_t)t.i1); t.p2 = ld3(sc.P + 3 * (size_t)t.i2);
    return t;
}
// dpdu of triangle.rs:548-574 (dpdv is needed only to detect the degenerate fallback)
PH_DEV void tri_dpdu(const DeviceScene& sc, const MeshRec& m, const TriVerts& t, f3& dpdu, f3& dpdv) {
    f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);  // get_uvs (:384-394)
    if (m.flags & PH_MESH_UV) {
        uv0 = mk2(sc.UV[2 * (size_t)t.i0], sc.UV[2 * (size_t)t.i0 + 1]);
        uv1 = mk2(sc.UV[2 * (size_t)t.i1], sc.UV[2 * (size_t)t.i1 + 1]);
        uv2 = mk2(sc.UV[2 * (size_t)t.i2], sc.UV[2 * (size_t)t.i2 + 1]);
    }
    f2 duv02 = mk2(uv0.x - uv2.x, uv0.y - uv2.y), duv12 = mk2(uv1.x - uv2.x, uv1.y - uv2.y);
    f3 dp02 = t.p0 - t.p2, dp12 = t.p1 - t.p2;
    float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
    bool degenerate_uv = fabsf(determinant) < 1e-8f;
    dpdu = mk3(0.0f, 0.0f, 0.0f); dpdv = mk3(0.0f, 0.0f, 0.0f);
    if (!degenerate_uv) {
        float invdet = ph_div(1.0f, determinant);
        dpdu = (duv12.y * dp02 - duv02.y * dp12) * invdet;
        dpdv = (-duv12.x * dp02 + duv02.x * dp12) * invdet;
    }
    if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
        f3 ng = cross(t.p2 - t.p0, t.p1 - t.p0);
        // ng == 0 is the "bogus" case, rejected during traversal (PH_TRI_BOGUS); never reached for an accepted hit
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
}
// tail of Triangle::intersect (triangle.rs:576-724) + Hit::new (interaction/mod.rs:137-156)
PH_DEV SurfHit make_surface_hit_tv(const DeviceScene& sc, const MeshRec& m, const TriVerts& t, f3 rd, float time, uint32_t prim, float b0, float b1, float b2);
PH_DEV SurfHit make_surface_hit(const DeviceScene& sc, f3 rd, float time, uint32_t prim, float b0, float b1, float b2) {
    const MeshRec m = sc.meshes[sc.tri_mesh[prim]];
    const TriVerts t = load_tri(sc, prim);
    return make_surface_hit_tv(sc, m, t, rd, time, prim, b0, b1, b2);
}
// Same, starting from the leaf-ordered TriRec the traversal kernel reported (hit.pad[0]): positions, primitive id and mesh id
// arrive in ONE 48-byte fetch instead of the tri_mesh -> meshes and idx -> P dependent chains.
PH_DEV SurfHit make_surface_hit_rec(const DeviceScene& sc, f3 rd, float time, uint32_t tri_index, float b0, float b1, float b2, MeshRec& m_out) {
    const float4* tp = reinterpret_cast<const float4*>(sc.tris + tri_index);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const uint32_t prim = __float_as_uint(a.w);
    const MeshRec m = sc.meshes[__float_as_uint(c.w)];
    TriVerts t;
    t.p0 = mk3(a.x, a.y, a.z); t.p1 = mk3(b.x, b.y, b.z); t.p2 = mk3(c.x, c.y, c.z);
    t.i0 = t.i1 = t.i2 = 0;
    if (m.flags & (PH_MESH_N | PH_MESH_S | PH_MESH_UV)) { t.i0 = sc.idx[3 * prim]; t.i1 = sc.idx[3 * prim + 1]; t.i2 = sc.idx[3 * prim + 2]; }
    m_out = m;
    return make_surface_hit_tv(sc, m, t, rd, time, prim, b0, b1, b2);
}
PH_DEV SurfHit make_surface_hit_tv(const DeviceScene& sc, const MeshRec& m, const TriVerts& t, f3 rd, float time, uint32_t prim, float b0, float b1, float b2) {
    SurfHit si;
    si.prim = prim; si.time = time;
    f3 dpdu, dpdv;
    tri_dpdu(sc, m, t, dpdu, dpdv);
    f3 dp02 = t.p0 - t.p2, dp12 = t.p1 - t.p2;
    float xs = fabsf(b0 * t.p0.x) + fabsf(b1 * t.p1.x) + fabsf(b2 * t.p2.x);
    float ys = fabsf(b0 * t.p0.y) + fabsf(b1 * t.p1.y) + fabsf(b2 * t.p2.y);
    float zs = fabsf(b0 * t.p0.z) + fabsf(b1 * t.p1.z) + fabsf(b2 * t.p2.z);
    si.p_error = kGamma7 * mk3(xs, ys, zs);
    si.p = b0 * t.p0 + b1 * t.p1 + b2 * t.p2;
    f3 wo = -rd;
    float l2 = length_squared(wo);
    si.wo = (l2 == 0.0f) ? wo : wo / ph_sqrt(l2);
    si.n = normalize(cross(dp02, dp12));
    const bool rev = (m.flags & PH_MESH_REV) != 0, swp = (m.flags & PH_MESH_SWAP) != 0;
    if (rev != swp) si.n = -si.n;
    si.ns = si.n; si.dpdu_s = dpdu;
    if (m.flags & (PH_MESH_N | PH_MESH_S)) {  // :631-721
        f3 ns;
        if (m.flags & PH_MESH_N) {
            f3 ns2 = b0 * ld3(sc.N + 3 * (size_t)t.i0) + b1 * ld3(sc.N + 3 * (size_t)t.i1) + b2 * ld3(sc.N + 3 * (size_t)t.i2);
            ns = length_squared(ns2) > 0.0f ? normalize(ns2) : si.n;
        } else ns = si.n;
        f3 ss;
        if (m.flags & PH_MESH_S) {
            f3 ss2 = b0 * ld3(sc.S + 3 * (size_t)t.i0) + b1 * ld3(sc.S + 3 * (size_t)t.i1) + b2 * ld3(sc.S + 3 * (size_t)t.i2);
            ss = length_squared(ss2) > 0.0f ? normalize(ss2) : normalize(dpdu);
        } else ss = normalize(dpdu);
        f3 ts = cross(ss, ns);
        if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, ss, ts);
        if (rev) ts = -ts;
        si.ns = normalize(cross(ss, ts));  // set_shading_geometry(.., true) (surface_interaction.rs:152-173)
        si.n = face_forward(si.n, si.ns);
        si.dpdu_s = ss;
    }
    return si;
}

// Ray::offset_origin (core/src/geometry/ray.rs:107-127)
PH_DEV f3 offset_origin(f3 p, f3 p_error, f3 n, f3 w) {
    float d = dot(vabs(n), p_error);
    f3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    f3 po = p + offset;
    if (offset.x > 0.0f) po.x = next_float_up(po.x); else if (offset.x < 0.0f) po.x = next_float_down(po.x);
    if (offset.y > 0.0f) po.y = next_float_up(po.y); else if (offset.y < 0.0f) po.y = next_float_down(po.y);
    if (offset.z > 0.0f) po.z = next_float_up(po.z); else if (offset.z < 0.0f) po.z = next_float_down(po.z);
    return po;
}
PH_DEV RayIn make_ray(f3 o, f3 d, float t_max, float time) {
    RayIn r; r.ox = o.x; r.oy = o.y; r.oz = o.z; r.t_max = t_max; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.time = time; return r;
}
PH_DEV RayIn spawn_ray(const SurfHit& h, f3 d) { return make_ray(offset_origin(h.p, h.p_error, h.n, d), d, kInf, h.time); }  // interaction/mod.rs:189-192
PH_DEV RayIn spawn_ray_to_hit(const SurfHit& h, f3 hp, f3 hperr, f3 hn) {  // interaction/mod.rs:212-223
    f3 origin = offset_origin(h.p, h.p_error, h.n, hp - h.p);
    f3 target = offset_origin(hp, hperr, hn, origin - hp);
    return make_ray(origin, target - origin, 1.0f - kShadowEps, h.time);
}

// =============================== matte BSDF =========================================================================
// BSDF::new (core/src/reflection/bsdf.rs:100-116) with one LambertianReflection / OrenNayar lobe (materials/src/matte.rs:47-76)
struct Bsdf {
    f3 ns, ng, ss, ts;
    spec r;
    float a, b;
    bool has_bxdf, oren;
};
PH_DEV Bsdf make_bsdf(const DeviceScene& sc, const SurfHit& si, uint32_t material) {
    const MaterialRec m = sc.materials[material];
    Bsdf b;
    b.ns = si.ns; b.ng = si.n; b.ss = normalize(si.dpdu_s); b.ts = cross(b.ns, b.ss);
    b.r = mks(m.kd[0], m.kd[1], m.kd[2]); b.has_bxdf = m.has_bxdf != 0; b.oren = m.sigma != 0.0f; b.a = m.a; b.b = m.b;
    return b;
}
PH_DEV f3 w2l(const Bsdf& b, f3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }  // bsdf.rs:118-120
PH_DEV f3 l2w(const Bsdf& b, f3 v) {                                                          // bsdf.rs:122-131
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
// core/src/reflection/common.rs
PH_DEV float sin2_theta(f3 w) { return pmaxf(0.0f, 1.0f - w.z * w.z); }
PH_DEV float sin_theta(f3 w) { return ph_sqrt(sin2_theta(w)); }
PH_DEV float cos_phi(f3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : pclampf(ph_div(w.x, s), -1.0f, 1.0f); }
PH_DEV float sin_phi(f3 w) { float s = sin_theta(w); return s == 0.0f ? 0.0f : pclampf(ph_div(w.y, s), -1.0f, 1.0f); }
PH_DEV spec bxdf_f(const Bsdf& b, f3 wo, f3 wi) {
    if (!b.oren) return b.r * kInvPi;  // lambertian_reflection.rs:38-40
    float sin_i = sin_theta(wi), sin_o = sin_theta(wo), max_cos = 0.0f;  // oren_nayar.rs:46-72
    if (sin_i > 1e-4f && sin_o > 1e-4f) {
        float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
        max_cos = pmaxf(0.0f, d_cos);
    }
    float aco = pabs(wo.z), aci = pabs(wi.z), sin_alpha, tan_beta;
    if (aci > aco) { sin_alpha = sin_o; tan_beta = ph_div(sin_i, aci); }
    else { sin_alpha = sin_i; tan_beta = ph_div(sin_o, aco); }
    return b.r * kInvPi * (b.a + b.b * max_cos * sin_alpha * tan_beta);
}
PH_DEV float bxdf_pdf(f3 wo, f3 wi) { return (wo.z * wi.z > 0.0f) ? pabs(wi.z) * kInvPi : 0.0f; }  // reflection/mod.rs:160-166
PH_DEV spec bsdf_f(const Bsdf& b, f3 wo_w, f3 wi_w) {  // bsdf.rs:133-158
    f3 wi = w2l(b, wi_w), wo = w2l(b, wo_w);
    if (wo.z == 0.0f) return mks1(0.0f);
    bool reflect = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
    spec f = mks1(0.0f);
    if (b.has_bxdf && reflect) f = f + bxdf_f(b, wo, wi);
    return f;
}
PH_DEV float bsdf_pdf(const Bsdf& b, f3 wo_w, f3 wi_w) {  // bsdf.rs:331-356
    if (!b.has_bxdf) return 0.0f;
    f3 wo = w2l(b, wo_w), wi = w2l(b, wi_w);
    if (wo.z == 0.0f) return 0.0f;
    float pdf = 0.0f;
    pdf += bxdf_pdf(wo, wi);
    return ph_div(pdf, 1.0f);
}
// BSDF::sample_f (bsdf.rs:194-292) for the single-lobe case; a failed sample is BxDFSample::default() (zeros)
PH_DEV void bsdf_sample_f(const Bsdf& b, f3 wo_w, f2 u, spec& f_out, float& pdf_out, f3& wi_out) {
    f_out = mks1(0.0f); pdf_out = 0.0f; wi_out = mk3(0.0f, 0.0f, 0.0f);
    if (!b.has_bxdf) return;
    // comp = min(floor(u0 * 1) as usize, 0) = 0
    f2 ur = mk2(pminf(u.x * 1.0f - 0.0f, kOneMinusEps), u.y);
    f3 wo = w2l(b, wo_w);
    if (wo.z == 0.0f) return;
    f3 wi = cosine_sample_hemisphere(ur);  // reflection/mod.rs:132-141
    if (wo.z < 0.0f) wi.z *= -1.0f;
    float pdf = bxdf_pdf(wo, wi);
    if (pdf == 0.0f) return;
    f3 wi_w = l2w(b, wi);
    bool reflect = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
    spec f = mks1(0.0f);
    if (reflect) f = f + bxdf_f(b, wo, wi);
    f_out = f; pdf_out = pdf; wi_out = wi_w;
}

// =============================== lights ==============================================================================
// constant InfiniteAreaLight: MIPMap::triangle on the single texel (core/src/mipmap/mod.rs:293-311)
PH_DEV float env_lookup1(float tx, float ds, float dt) { return tx * (1.0f - ds) * (1.0f - dt) + tx * (1.0f - ds) * dt + tx * ds * (1.0f - dt) + tx * ds * dt; }
PH_DEV spec infinite_lookup(const LightRec& l, f2 st) {
    float s = st.x * 1.0f - 0.5f, t = st.y * 1.0f - 0.5f;
    float s0 = floorf(s), t0 = floorf(t);
    float ds = s - s0, dt = t - t0;
    return mks(env_lookup1(l.L[0], ds, dt), env_lookup1(l.L[1], ds, dt), env_lookup1(l.L[2], ds, dt));
}
PH_DEV f3 xf_vec(const float* m, f3 v) {  // transform_vector (transform.rs:373-380) on a 3x4 row-major block
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// ---- object instances: Transform::transform_surface_interaction (core/src/geometry/transform.rs:566-590) ------------------------
PH_DEV f3 xf_normal(const float* m_inv, f3 n) {  // transform_normal (transform.rs:441-448): transpose of the inverse
    return mk3(m_inv[0] * n.x + m_inv[4] * n.y + m_inv[8] * n.z, m_inv[1] * n.x + m_inv[5] * n.y + m_inv[9] * n.z, m_inv[2] * n.x + m_inv[6] * n.y + m_inv[10] * n.z);
}
// transform_point_with_abs_error (transform.rs:338-370); returns the point, writes the new absolute error
PH_DEV f3 xf_point_abs_err(const float* m, f3 p, f3 pe, f3& err) {
    const float x = p.x, y = p.y, z = p.z;
    const float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    const float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    const float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    const float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    const float g3 = kGamma3;
    err = mk3((g3 + 1.0f) * (pabs(m[0]) * pe.x + pabs(m[1]) * pe.y + pabs(m[2]) * pe.z) + g3 * (pabs(m[0] * x) + pabs(m[1] * y) + pabs(m[2] * z) + pabs(m[3])),
              (g3 + 1.0f) * (pabs(m[4]) * pe.x + pabs(m[5]) * pe.y + pabs(m[6]) * pe.z) + g3 * (pabs(m[4] * x) + pabs(m[5] * y) + pabs(m[6] * z) + pabs(m[7])),
              (g3 + 1.0f) * (pabs(m[8]) * pe.x + pabs(m[9]) * pe.y + pabs(m[10]) * pe.z) + g3 * (pabs(m[8] * x) + pabs(m[9] * y) + pabs(m[10] * z) + pabs(m[11])));
    return (wp == 1.0f) ? mk3(xp, yp, zp) : mk3(xp, yp, zp) / wp;
}
// The interaction TransformedPrimitive::intersect hands back (transformed_primitive.rs:51-73): the triangle met the instance-space
// ray, then everything the integrator reads is carried to world space.  inst = instance number + 1 (HitOut pad[1]), 0 = none.
PH_DEV SurfHit make_surface_hit_any(const DeviceScene& sc, f3 rd_world, float time, uint32_t tri_index, uint32_t inst, float b0, float b1, float b2, MeshRec& m_out) {
    if (inst == 0u) return make_surface_hit_rec(sc, rd_world, time, tri_index, b0, b1, b2, m_out);
    const InstRec& I = sc.instances[inst - 1u];
    SurfHit si = make_surface_hit_rec(sc, xf_vec(I.w2i, rd_world), time, tri_index, b0, b1, b2, m_out);
    if (I.flags & PH_INST_IDENTITY) return si;
    f3 pe;
    si.p = xf_point_abs_err(I.i2w, si.p, si.p_error, pe); si.p_error = pe;
    si.wo = normalize(xf_vec(I.i2w, si.wo));
    si.n = normalize(xf_normal(I.w2i, si.n));
    si.ns = normalize(xf_normal(I.w2i, si.ns));
    si.ns = face_forward(si.ns, si.n);
    si.dpdu_s = xf_vec(I.i2w, si.dpdu_s);
    return si;
}
PH_DEV spec area_L(const LightRec& l, f3 n, f3 w) {  // DiffuseAreaLight::l (lights/src/diffuse.rs:220-226)
    return (l.two_sided || dot(n, w) > 0.0f) ? mks(l.L[0], l.L[1], l.L[2]) : mks1(0.0f);
}
// Radiance-map variants of the three infinite-light operations, out of line (texture.h): `dsc` = DeviceScene::self
static __device__ __noinline__ spec envmap_lookup(const DeviceScene* dsc, uint32_t mip, float s, float t);
static __device__ __noinline__ float envmap_sample(const DeviceScene* dsc, uint32_t dist_off, uint32_t dw, uint32_t dh, float u0, float u1, float* d0, float* d1);
static __device__ __noinline__ float envmap_pdf(const DeviceScene* dsc, uint32_t dist_off, uint32_t dw, uint32_t dh, float px, float py);
// MAP = false compiles the radiance-map branches out: the texture-free shade kernels (the headline workload) must not pay for the out-of-line calls
template <bool MAP = true> PH_DEV spec light_le(const DeviceScene& sc, const LightRec& l, f3 ray_d) {  // Light::le: InfiniteAreaLight (infinite.rs:188-195); zero for the others
    if (l.type != PH_L_INFINITE) return mks1(0.0f);
    f3 w = normalize(xf_vec(l.w2l, ray_d));
    const float s = spherical_phi(w) * kInvTwoPi, t = spherical_theta(w) * kInvPi;
    if (MAP && l.map_mip1) return envmap_lookup(sc.self, l.map_mip1 - 1u, s, t);
    return infinite_lookup(l, mk2(s, t));
}
// Distribution1D::sample_continuous on a 2-entry function (distribution_1d.rs:55-79)
PH_DEV float dist2_sample_continuous(const float* func, const float* cdf, float func_int, float u, float& pdf, uint32_t& off) {
    uint32_t offset = find_interval_cdf(cdf, 3, u);
    float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du = ph_div(du, cdf[offset + 1] - cdf[offset]);
    pdf = func_int > 0.0f ? ph_div(func[offset], func_int) : 0.0f;
    off = offset;
    return ph_div((float)offset + du, 2.0f);
}

struct LiSample { f3 wi; float pdf; spec value; f3 vp, vperr, vn; bool valid; };
// Triangle::area-light sampling record, shared by sample_li
template <bool MAP = true> PH_DEV LiSample light_sample_li(const DeviceScene& sc, const LightRec& l, const SurfHit& hit, f2 u) {
    LiSample r;
    r.valid = false; r.pdf = 0.0f; r.wi = mk3(0, 0, 0); r.value = mks1(0.0f);
    r.vp = mk3(0, 0, 0); r.vperr = mk3(0, 0, 0); r.vn = mk3(0, 0, 0);
    if (l.type == PH_L_INFINITE) {  // infinite.rs:133-173
        float d0, d1, map_pdf;
        if (MAP && l.map_mip1) map_pdf = envmap_sample(sc.self, l.dist_off, l.dw, l.dh, u.x, u.y, &d0, &d1);
        else {
            float pdf1, pdf0; uint32_t v, dummy;
            d1 = dist2_sample_continuous(l.marg_func, l.marg_cdf, l.marg_int, u.y, pdf1, v);
            d0 = dist2_sample_continuous(l.cond_func + 2 * v, l.cond_cdf + 3 * v, l.cond_int[v], u.x, pdf0, dummy);
            map_pdf = pdf0 * pdf1;
        }
        if (map_pdf == 0.0f) return r;
        float theta = d1 * kPi, phi = d0 * kTwoPi;
        float cos_theta, sin_theta, sin_phi_, cos_phi_;
        d_sincos(theta, sin_theta, cos_theta);
        d_sincos(phi, sin_phi_, cos_phi_);
        r.wi = xf_vec(l.l2w, mk3(sin_theta * cos_phi_, sin_theta * sin_phi_, cos_theta));
        r.pdf = ph_div(map_pdf, kTwoPi * kPi * sin_theta);
        if (sin_theta == 0.0f) r.pdf = 0.0f;
        r.vp = hit.p + r.wi * (2.0f * sc.world_radius);
        r.value = (MAP && l.map_mip1) ? envmap_lookup(sc.self, l.map_mip1 - 1u, d0, d1) : infinite_lookup(l, mk2(d0, d1));
        r.valid = true;
    } else if (l.type == PH_L_DISTANT) {  // distant.rs:87-96
        f3 w = mk3(l.v[0], l.v[1], l.v[2]);
        r.wi = w; r.pdf = 1.0f; r.vp = hit.p + w * (2.0f * sc.world_radius); r.value = mks(l.L[0], l.L[1], l.L[2]); r.valid = true;
    } else if (l.type == PH_L_SPOT) {  // spot.rs:75-84, falloff :52-66
        f3 pl = mk3(l.v[0], l.v[1], l.v[2]);
        r.wi = normalize(pl - hit.p); r.pdf = 1.0f; r.vp = pl;
        const f3 wl = normalize(xf_vec(l.w2l, -r.wi));
        const float cos_theta = wl.z;
        float fall;
        if (cos_theta < l.cos_total_width) fall = 0.0f;
        else if (cos_theta >= l.cos_falloff_start) fall = 1.0f;
        else { const float delta = ph_div(cos_theta - l.cos_total_width, l.cos_falloff_start - l.cos_total_width); fall = (delta * delta) * (delta * delta); }
        r.value = mks(l.L[0], l.L[1], l.L[2]) * fall / distance_squared(pl, hit.p); r.valid = true;
    } else if (l.type == PH_L_PROJECTION) {  // projection.rs:180-191, projection() :145-168
        f3 pl = mk3(l.v[0], l.v[1], l.v[2]);
        r.wi = normalize(pl - hit.p); r.pdf = 1.0f; r.vp = pl;
        spec pr = mks1(0.0f);
        const f3 wl = xf_vec(l.w2l, -r.wi);
        if (!(wl.z < 1e-3f)) {
            const float* m = l.proj;   // Transform::transform_point (transform.rs:288-302)
            const float xp = m[0] * wl.x + m[1] * wl.y + m[2] * wl.z + m[3], yp = m[4] * wl.x + m[5] * wl.y + m[6] * wl.z + m[7];
            const float zp = m[8] * wl.x + m[9] * wl.y + m[10] * wl.z + m[11], wp = m[12] * wl.x + m[13] * wl.y + m[14] * wl.z + m[15];
            const f3 pp = (wp == 1.0f) ? mk3(xp, yp, zp) : mk3(xp, yp, zp) / wp;
            if (pp.x >= l.screen[0] && pp.x <= l.screen[1] && pp.y >= l.screen[2] && pp.y <= l.screen[3]) {
                if (!(MAP && l.map_mip1)) pr = mks1(1.0f);
                else {
                    float ox = pp.x - l.screen[0], oy = pp.y - l.screen[2];   // Bounds2::offset (bounds2.rs:161-173)
                    if (l.screen[1] > l.screen[0]) ox = ph_div(ox, l.screen[1] - l.screen[0]);
                    if (l.screen[3] > l.screen[2]) oy = ph_div(oy, l.screen[3] - l.screen[2]);
                    pr = envmap_lookup(sc.self, l.map_mip1 - 1u, ox, oy);
                }
            }
        }
        r.value = mks(l.L[0], l.L[1], l.L[2]) * pr / distance_squared(pl, hit.p); r.valid = true;
    } else if (l.type == PH_L_GONIO) {  // goniometric.rs:102-113, scale() :82-96
        f3 pl = mk3(l.v[0], l.v[1], l.v[2]);
        r.wi = normalize(pl - hit.p); r.pdf = 1.0f; r.vp = pl;
        f3 wp = normalize(xf_vec(l.w2l, -r.wi));
        { const float t = wp.y; wp.y = wp.z; wp.z = t; }
        const float theta = spherical_theta(wp), phi = spherical_phi(wp);
        const spec scl = (MAP && l.map_mip1) ? envmap_lookup(sc.self, l.map_mip1 - 1u, phi * kInvTwoPi, theta * kInvPi) : mks1(1.0f);
        r.value = mks(l.L[0], l.L[1], l.L[2]) * scl / distance_squared(pl, hit.p); r.valid = true;
    } else if (l.type == PH_L_POINT) {  // point.rs:83-93
        f3 pl = mk3(l.v[0], l.v[1], l.v[2]);
        r.wi = normalize(pl - hit.p); r.pdf = 1.0f; r.vp = pl;
        r.value = mks(l.L[0], l.L[1], l.L[2]) / distance_squared(pl, hit.p); r.valid = true;
    } else {  // DiffuseAreaLight::sample_li (diffuse.rs:114-129) -> sample_solid_angle (shape.rs:64-84) -> Triangle::sample (triangle.rs:918-949)
        const MeshRec m = sc.meshes[sc.tri_mesh[l.prim]];
        const TriVerts t = load_tri(sc, l.prim);
        f2 b = uniform_sample_triangle(u);
        f3 p = b.x * t.p0 + b.y * t.p1 + (1.0f - b.x - b.y) * t.p2;
        f3 n = normalize(cross(t.p1 - t.p0, t.p2 - t.p0));
        if (m.flags & PH_MESH_N) {
            f3 ns = b.x * ld3(sc.N + 3 * (size_t)t.i0) + b.y * ld3(sc.N + 3 * (size_t)t.i1) + (1.0f - b.x - b.y) * ld3(sc.N + 3 * (size_t)t.i2);
            n = face_forward(n, ns);
        } else if (((m.flags & PH_MESH_REV) != 0) != ((m.flags & PH_MESH_SWAP) != 0)) n = n * -1.0f;
        f3 p_abs_sum = vabs(b.x * t.p0) + vabs(b.y * t.p1) + vabs((1.0f - b.x - b.y) * t.p2);
        f3 p_error = kGamma6 * mk3(p_abs_sum.x, p_abs_sum.y, p_abs_sum.z);
        float pdf = ph_div(1.0f, l.area);
        f3 wi = p - hit.p;
        if (length_squared(wi) == 0.0f) pdf = 0.0f;
        else {
            wi = normalize(wi);
            pdf *= ph_div(distance_squared(hit.p, p), abs_dot(n, -wi));
            if (__builtin_isinf(pdf)) pdf = 0.0f;
        }
        f3 wi2 = p - hit.p;
        float l2 = length_squared(wi2);
        if (pdf == 0.0f || l2 == 0.0f) return r;
        wi2 = wi2 / ph_sqrt(l2);
        r.wi = wi2; r.pdf = pdf; r.value = area_L(l, n, -wi2); r.vp = p; r.vperr = p_error; r.vn = n; r.valid = true;
    }
    return r;
}
template <bool MAP = true> PH_DEV float light_pdf_li(const DeviceScene& sc, const LightRec& l, const SurfHit& hit, f3 wi) {
    if (l.type == PH_L_INFINITE) {  // infinite.rs:201-211 + Distribution2D::pdf (distribution_2d.rs:51-65)
        f3 w = xf_vec(l.w2l, wi);
        float theta = spherical_theta(w), phi = spherical_phi(w), sin_theta = d_sin(theta);
        if (sin_theta == 0.0f) return 0.0f;
        if (MAP && l.map_mip1) return ph_div(envmap_pdf(sc.self, l.dist_off, l.dw, l.dh, phi * kInvTwoPi, theta * kInvPi), kTwoPi * kPi * sin_theta);
        uint32_t iu = f2u_sat(phi * kInvTwoPi * 2.0f), iv = f2u_sat(theta * kInvPi * 2.0f);
        if (iu > 1) iu = 1;
        if (iv > 1) iv = 1;
        return ph_div(ph_div(l.cond_func[2 * iv + iu], l.marg_int), kTwoPi * kPi * sin_theta);
    }
    if (l.type == PH_L_AREA) {  // Shape::pdf_solid_angle (core/src/geometry/shape.rs:86-107): one Triangle::intersect, no BVH
        RayIn ray = spawn_ray(hit, wi);
        RayState rs;
        ray_setup(rs, ray);
        const TriVerts t = load_tri(sc, l.prim);
        float tt, b0, b1, b2;
        if (!tri_test(rs, t.p0, t.p1, t.p2, tt, b0, b1, b2)) return 0.0f;
        if (sc.tri_flags[l.prim] & PH_TRI_BOGUS) return 0.0f;  // test_alpha = false: only the degenerate rejection applies
        SurfHit lh = make_surface_hit(sc, wi, hit.time, l.prim, b0, b1, b2);
        float pdf = ph_div(distance_squared(hit.p, lh.p), abs_dot(lh.n, -wi) * l.area);
        return __builtin_isinf(pdf) ? 0.0f : pdf;
    }
    return 0.0f;
}

}  // namespace ph
