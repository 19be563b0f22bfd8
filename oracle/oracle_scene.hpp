// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// Scene data, BVH build + traversal, watertight triangle test.
//   accelerators/src/bvh/{mod,common,sah}.rs, core/src/geometry/bounds3.rs, shapes/src/triangle.rs,
//   core/src/primitives/geometric_primitive.rs
// Third-party code on this path that is absent from /root/reference (Cargo.toml:33,36; no Cargo.lock):
//   itertools = "0.13"  `partition` (sah.rs:229,354) — restated below from its published algorithm;
//   order-stat = "0.1"  `kth_by`    (sah.rs:275)     — under SAH only ever called with 2 elements, where any
//                                                       correct selection gives [smaller, larger].
#pragma once
#include "oracle_math.hpp"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <vector>
#include "oracle_texture.hpp"

namespace orc {

struct Ray {  // core/src/geometry/ray.rs:10-28 (medium out of scope)
    V3 o, d;
    Float t_max, time;
    bool has_diff = false;          // ray.differentials: Some only for camera rays (spawn_ray never sets them)
    V3 rx_o, ry_o, rx_d, ry_d;
    Ray() : t_max(INF), time(0) {}
    Ray(V3 o_, V3 d_, Float tm, Float ti) : o(o_), d(d_), t_max(tm), time(ti) {}
};

// A material with constant textures always produces the same list of BxDFs (compute_scattering_functions only evaluates
// textures), so the list is made once when the material is created.  core/src/reflection/*, materials/src/*.rs
enum { BX_REFL = 1, BX_TRANS = 2, BX_DIFF = 4, BX_GLOSSY = 8, BX_SPEC = 16, BX_ALL = 31 };  // BxDFType (bsdf.rs:10-20)
enum LobeKind { LK_LAMBERT = 0, LK_OREN = 1, LK_SPEC_R = 2, LK_SPEC_T = 3, LK_FRESNEL_SPEC = 4, LK_MICRO_R = 5, LK_MICRO_T = 6, LK_FRESNEL_BLEND = 7, LK_LAMBERT_T = 8 };
enum FresnelKind { FR_NOOP = 0, FR_DIEL = 1, FR_COND = 2 };
struct Lobe {
    int kind = LK_LAMBERT, type = BX_REFL | BX_DIFF, fresnel = FR_NOOP;
    Spec r, t;                 // reflectance / transmittance scale
    Float a = 0, b = 0;        // Oren-Nayar A, B
    Float ax = 0, ay = 0;      // Trowbridge-Reitz alpha (already max(0.001, .))
    Float eta_a = 1, eta_b = 1;  // dielectric indices (Fresnel eta_i/eta_t, or etaA/etaB of the transmission lobes)
    Spec c_eta_i, c_eta_t, c_k;  // conductor Fresnel
    int n_scale = 0; Spec scale[2];  // ScaledBxDF wrappers of MixMaterial, innermost first (scaled_bxdf.rs)
    int r_tex = -1, t_tex = -1;      // this colour is a texture evaluated per hit (set_material_texture)
    bool has_pre = false; Spec pre;  // ... multiplied by `pre` (UberMaterial's opacity, uber.rs:133)
    bool pre_raw_test = false;       // TranslucentMaterial: `if !kd.is_black() { add(r * kd) }` tests the texel BEFORE the product (translucent.rs:77-84, :87)
    int ax_tex = -1, ay_tex = -1; bool remap = false;  // the microfacet roughness is a float texture (plastic.rs:71-76, uber.rs:140-153, substrate.rs:63-71, metal.rs:69-83)
    int sigma_tex = -1;              // MatteMaterial's sigma is a float texture (matte.rs:64-70)
    int eta_tex = -1, k_tex = -1;    // MetalMaterial's eta / k are textures (metal.rs:121-125): evaluated per hit, NOT clamped
    int amt_side = 0, amt_level = 0; // MixMaterial with an `amount` texture (mix.rs:59-60): scale[amt_level] is s1 (side 1) or s2 = clamp(1 - s1) (side 2) of the hit
    int alt = 0;                     // GlassMaterial with roughness textures (glass.rs:110-141): 1 = the lobe of a hit where urough == vrough == 0, 2 = a lobe of the other hits
    Float ur_raw = 0, vr_raw = 0;    // ... the constant roughnesses BEFORE remapping (what `is_specular` compares with 0)
    int pre_mode = 0;                // UberMaterial with an opacity texture (uber.rs:126-160): 3 = colour = op(hit) * (texel, or the constant kept in `pre`), 4 = colour = clamp(1 - op(hit));
                                     // 5 = TranslucentMaterial with a reflect / transmit texture (translucent.rs:70-98): colour = reflect-or-transmit(hit) * (texel, or the constant Kd / Ks kept in `pre`), both tested for black
};
struct Material {
    Spec kd; Float sigma; std::vector<Lobe> lobes; Float bsdf_eta = 1.0f; bool general = false; bool none = false;  // none: Material "none" / "" -> no BSDF at all
    int bump_tex = -1;      // Material::bump's displacement texture (material.rs:62-101), any material
    bool textured = false;  // some lobe colour is a texture: the BSDF's lobe list is made per hit (compute_scattering_functions evaluates the textures there)
    int param_lobe[4] = {-1, -1, -1, -1}, param_field[4] = {0, 0, 0, 0};  // [Kd, Ks, Kr, Kt] -> lobe index / 0 = r, 1 = t
    int param_lobe2[4] = {-1, -1, -1, -1}, param_field2[4] = {0, 0, 0, 0};  // a second lobe fed by the same parameter (translucent: the reflection and the transmission lobe)
    int rough_lobe = -1, rough_lobe2 = -1; bool rough_remap = false;   // the lobe(s) that own the Trowbridge-Reitz distribution (set_material_float_texture)
    bool has_pre = false; Spec pre;
    int index_tex = -1;     // GlassMaterial / UberMaterial: `index` is a float texture, evaluated at every hit (glass.rs:102, uber.rs:128)
    int opacity_tex = -1;   // UberMaterial's opacity is a texture: which lobes a hit gets, their colours and BSDF::eta are decided per hit
    Float bsdf_eta_alt = 1.0f;  // ... BSDF::eta where the pass-through lobe is NOT added (uber.rs:136)
    int amount_tex = -1;    // MixMaterial's amount is a texture
    int mix_n1 = 0;         // MixMaterial: how many of the lobes come from the first material
    // what the material was created from, for the setters that have to rebuild the lobe list (opacity / glass roughness textures)
    int refl_tex = -1, trans_tex = -1;   // TranslucentMaterial: reflect / transmit are textures (translucent.rs:70-71); where both are black a hit has NO BSDF (:72-74)
    bool rt_mode = false;                // ... the lobe list is in the per-hit form (pre_mode 5 lobes)
    int made_as = 0;        // 0 other, 1 uber, 2 glass, 3 metal, 4 mix, 5 translucent
    Spec raw_k[4]; Float raw_eta = 1.5f, raw_ur = 0, raw_vr = 0; bool raw_remap = false; bool rebuilt = false;
};

enum LightType { L_INFINITE = 0, L_DISTANT = 1, L_POINT = 2, L_AREA = 3, L_SPOT = 4, L_PROJECTION = 5, L_GONIO = 6 };
struct Light {
    int type;
    Spec L;             // infinite: lrgb; distant: emitted radiance; point: intensity; area: l_emit
    Transform l2w;      // infinite only
    V3 w_light;         // distant
    V3 p_light;         // point, spot
    Float cos_total_width = 0, cos_falloff_start = 0;  // spot (lights/src/spot.rs:24-25); l2w holds its light_to_world
    int two_sided;      // area
    uint32_t prim;      // area: global triangle index bound to this light
    int sphere = -1;    // area light whose shape is scene.spheres[sphere] (ORACLE ONLY, for the reference's lights/diffuse.pbrt); prim is then the sphere's primitive slot
    Float area;         // area: Triangle::area()
    // infinite: 2x2 scalar image distribution (lights/src/infinite.rs:326-369, sampling/distribution_2d.rs)
    Float cond_func[2][2], cond_cdf[2][3], cond_int[2];
    Float marg_func[2], marg_cdf[3], marg_int;
    // infinite with a radiance map (mapname): the MIPMap (scene.mipmaps[map_mip], built unflipped) and the Distribution2D over its 2w x 2h scalar image
    int map_mip = -1; int dw = 0, dh = 0;
    // projection (lights/src/projection.rs): light_projection, screen_bounds {xmin, xmax, ymin, ymax}; cos_total_width above; map_mip = its image (or -1).  goniometric: map_mip only
    Transform light_projection; Float screen[4] = {0, 0, 0, 0};
    std::vector<Float> d_cond_func, d_cond_cdf, d_cond_int, d_marg_func, d_marg_cdf; Float d_marg_int = 0;
};

struct Mesh {
    uint32_t vert_base, tri_base, n_verts, n_tris;
    bool has_n, has_s, has_uv;
    uint32_t material;
    int32_t first_light;
    bool reverse_orientation, swaps_handedness;
    Float alpha, shadow_alpha;
    int alpha_tex = -1, shadow_alpha_tex = -1;   // float textures instead of the constants (triangle.rs:587-607, 868-898)
    int sphere = -1;        // >= 0: this record stands for ONE Sphere (scene.spheres[sphere]) that occupies one slot of the primitive list; no vertices
    int hyper = -1;         // the same for a Hyperboloid (scene.hyperboloids[hyper])
    int quadric = -1;       // ... and for a Cylinder / Cone / Paraboloid / Disk (scene.quadrics[quadric])
};

// Sphere (shapes/src/sphere.rs:10-57) — ORACLE ONLY: BASELINE.json's configs[0] (scenes/shapes/sphere.pbrt) is "CPU reference only (plumbing)", the product renders triangles
struct Sphere {
    Transform o2w, w2o;
    bool reverse_orientation = false, swaps_handedness = false;
    Float radius = 1, z_min = -1, z_max = 1, theta_min = 0, theta_max = 0, phi_max = 0;
    Sphere() {}
    Sphere(const Transform& o2w_, bool rev, Float radius_, Float zmin, Float zmax, Float phimax_deg) : o2w(o2w_), w2o(o2w_.inv()), reverse_orientation(rev), radius(radius_) {  // Sphere::new (:21-44)
        swaps_handedness = o2w.swaps_handedness();  // ShapeData::new
        z_min = pclamp(pmin(zmin, zmax), -radius, radius);
        z_max = pclamp(pmax(zmin, zmax), -radius, radius);
        theta_min = o_acos(pclamp(pmin(zmin, zmax) / radius, -1.0f, 1.0f));
        theta_max = o_acos(pclamp(pmax(zmin, zmax) / radius, -1.0f, 1.0f));
        phi_max = pclamp(phimax_deg, 0.0f, 360.0f) * (PI / 180.0f);  // f32::to_radians
    }
    Bounds3 object_bound() const { return Bounds3(V3(-radius, -radius, z_min)).union_p(V3(radius, radius, z_max)); }  // :53-58 (Bounds3::new of two sorted corners)
};

// Hyperboloid (shapes/src/hyperboloid.rs:24-110) — ORACLE ONLY, for the reference's textures/2d-mappings scene (image maps under the four 2-D mappings)
struct Hyperboloid {
    Transform o2w, w2o;
    bool reverse_orientation = false, swaps_handedness = false;
    V3 p1, p2; Float z_min = 0, z_max = 0, phi_max = 0, r_max = 0, ah = 0, ch = 0;
    Hyperboloid() {}
    Hyperboloid(const Transform& o2w_, bool rev, V3 point1, V3 point2, Float phimax_deg) : o2w(o2w_), w2o(o2w_.inv()), reverse_orientation(rev) {
        swaps_handedness = o2w.swaps_handedness();
        p1 = point1; p2 = point2;
        const Float radius1 = std::sqrt(p1.x * p1.x + p1.y * p1.y), radius2 = std::sqrt(p2.x * p2.x + p2.y * p2.y);
        r_max = pmax(radius1, radius2); z_min = pmin(p1.z, p2.z); z_max = pmax(p1.z, p2.z);
        if (p2.z == 0.0f) { V3 t = p1; p1 = p2; p2 = t; }
        V3 pp = p1;
        for (int count = 0;; count++) {  // :70-88
            pp = pp + 2.0f * (p2 - p1);
            const Float xy1 = pp.x * pp.x + pp.y * pp.y, xy2 = p2.x * p2.x + p2.y * p2.y;
            ah = (1.0f / xy1 - (pp.z * pp.z) / (xy1 * p2.z * p2.z)) / (1.0f - (xy2 * pp.z * pp.z) / (xy1 * p2.z * p2.z));
            ch = (ah * xy2 - 1.0f) / (p2.z * p2.z);
            if (std::isfinite(ah) || count > 100000) break;
        }
        phi_max = pclamp(phimax_deg, 0.0f, 360.0f) * (PI / 180.0f);
    }
    Bounds3 object_bound() const { return Bounds3(V3(-r_max, -r_max, z_min)).union_p(V3(r_max, r_max, z_max)); }
};

// Cylinder, Cone, Paraboloid, Disk (shapes/src/{cylinder,cone,paraboloid,disk}.rs) — ORACLE ONLY: the other four quadrics of the reference's six-shape texture scenes
enum QuadricKind { Q_CYLINDER = 0, Q_CONE = 1, Q_PARABOLOID = 2, Q_DISK = 3 };
struct Quadric {
    int kind = Q_CYLINDER;
    Transform o2w, w2o;
    bool reverse_orientation = false, swaps_handedness = false;
    Float radius = 1, z_min = 0, z_max = 1, height = 1, inner_radius = 0, phi_max = 0;
    Quadric() {}
    // a, b: cylinder / paraboloid (zmin, zmax); cone (height, -); disk (height, innerradius)
    Quadric(int kind_, const Transform& o2w_, bool rev, Float radius_, Float a, Float b, Float phimax_deg) : kind(kind_), o2w(o2w_), w2o(o2w_.inv()), reverse_orientation(rev), radius(radius_) {
        swaps_handedness = o2w.swaps_handedness();
        if (kind == Q_CYLINDER || kind == Q_PARABOLOID) { z_min = pmin(a, b); z_max = pmax(a, b); }
        else if (kind == Q_CONE) height = a;
        else { height = a; inner_radius = b; }
        phi_max = pclamp(phimax_deg, 0.0f, 360.0f) * (PI / 180.0f);
    }
    Bounds3 object_bound() const {
        if (kind == Q_CONE) return Bounds3(V3(-radius, -radius, 0.0f)).union_p(V3(radius, radius, height));
        if (kind == Q_DISK) return Bounds3(V3(-radius, -radius, height)).union_p(V3(radius, radius, height));
        return Bounds3(V3(-radius, -radius, z_min)).union_p(V3(radius, radius, z_max));
    }
};

struct LinearBVHNode {  // accelerators/src/bvh/common.rs:163-179 (32 bytes)
    Bounds3 bounds;
    uint32_t offset;
    uint16_t n_primitives;
    uint8_t axis, pad;
};

struct TriHit { Float t, b0, b1, b2; uint32_t inst = 0; V3 sp; Float sphi = 0; V3 sperr; Float sv = 0; };  // sp / sphi: a Sphere hit's refined object-space point and phi  // inst = instance number + 1 when the hit lies inside an ObjectInstance

// Object instancing (api/src/lib.rs:911-1000, core/src/primitives/transformed_primitive.rs).  An object is a contiguous range
// of the scene's triangles; when it is instanced the reference wraps its primitives in one aggregate (the BVH the Accelerator
// directive names) unless there is exactly one, which is used directly (lib.rs:953-971).
#define ORC_INST_BIT 0x80000000u
struct Object {
    uint32_t tri0 = 0, tri1 = 0;
    bool built = false;
    std::vector<LinearBVHNode> nodes; std::vector<uint32_t> ordered_prims;
    Bounds3 bound;
};
struct Instance { uint32_t object; Transform i2w; };

// Transform::transform_ray (core/src/geometry/transform.rs:451-476) incl. quirk B2 (t_max -= dt)
inline Ray transform_ray(const Transform& t, const Ray& r) {
    V3 o_error; V3 o = t.point_with_error(r.o, o_error);
    V3 d = t.vector(r.d);
    Float l2 = length_squared(d), t_max = r.t_max;
    if (l2 > 0.0f) { Float dt = dot(vabs(d), o_error) / l2; o = o + d * dt; t_max -= dt; }
    return Ray(o, d, t_max, r.time);
}
inline Bounds3 transform_bounds(const Transform& t, const Bounds3& b) {  // transform.rs:552-561
    return Bounds3(t.point(V3(b.pmin.x, b.pmin.y, b.pmin.z)))
        .union_p(t.point(V3(b.pmax.x, b.pmin.y, b.pmin.z))).union_p(t.point(V3(b.pmin.x, b.pmax.y, b.pmin.z)))
        .union_p(t.point(V3(b.pmin.x, b.pmin.y, b.pmax.z))).union_p(t.point(V3(b.pmin.x, b.pmax.y, b.pmax.z)))
        .union_p(t.point(V3(b.pmax.x, b.pmax.y, b.pmin.z))).union_p(t.point(V3(b.pmax.x, b.pmin.y, b.pmax.z)))
        .union_p(t.point(V3(b.pmax.x, b.pmax.y, b.pmax.z)));
}

struct TraversalStats { uint64_t nodes_visited = 0, tri_tests = 0, rays = 0; };

struct Scene {
    // geometry (all meshes concatenated)
    std::vector<V3> P, N, S;
    std::vector<V2> UV;
    std::vector<uint32_t> idx;        // 3 per triangle, already offset by vert_base
    std::vector<uint32_t> tri_mesh;   // mesh id per triangle
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<Texture> textures;    // float and spectrum textures share one id space
    std::vector<MipMap> mipmaps;
    std::vector<Light> lights;
    // DiffuseAreaLights of shapes inside an object definition: the reference keeps them on the primitives (their emission is seen where a path looks at the surface) but
    // never adds them to the scene's lights — "Area lights not supported with object instancing" (api/src/lib.rs:877-881).  Mesh::first_light = -2 - index for those.
    std::vector<Light> emission_only;
    std::vector<int> infinite_lights;
    std::vector<Sphere> spheres;      // oracle-only shapes; each owns one Mesh record and one primitive slot
    std::vector<Hyperboloid> hyperboloids;
    std::vector<Quadric> quadrics;
    // objects / instances; top_items = the scene's primitive list in directive order (triangle id, or ORC_INST_BIT | instance)
    std::vector<Object> objects;
    std::vector<Instance> instances;
    std::vector<uint32_t> top_items;
    int open_object = -1;
    // BVH
    std::vector<LinearBVHNode> nodes;
    std::vector<uint32_t> ordered_prims;
    int max_prims_in_node = 4;
    Bounds3 world_bound;
    V3 world_center;
    Float world_radius = 1.0f;

    size_t n_tris() const { return idx.size() / 3; }
    const Mesh& mesh_of(uint32_t prim) const { return meshes[tri_mesh[prim]]; }

    // ---- Triangle::world_bound (triangle.rs:427-431)
    Bounds3 tri_bound(uint32_t prim) const {
        if (mesh_of(prim).sphere >= 0) { const Sphere& sp = spheres[(size_t)mesh_of(prim).sphere]; return transform_bounds(sp.o2w, sp.object_bound()); }  // Shape::world_bound (shape.rs)
        if (mesh_of(prim).hyper >= 0) { const Hyperboloid& hy = hyperboloids[(size_t)mesh_of(prim).hyper]; return transform_bounds(hy.o2w, hy.object_bound()); }
        if (mesh_of(prim).quadric >= 0) { const Quadric& q = quadrics[(size_t)mesh_of(prim).quadric]; return transform_bounds(q.o2w, q.object_bound()); }
        return Bounds3(P[idx[3 * prim]]).union_p(P[idx[3 * prim + 1]]).union_p(P[idx[3 * prim + 2]]);
    }

    // ---- Sphere::intersect / intersect_p up to the accept decision (sphere.rs:59-171 / 243-330): the two run the same tests.  Fills t and the
    // refined object-space hit point + phi; the SurfaceInteraction is built from them by the renderer (sphere.rs:173-241).
    bool sphere_intersect(const Ray& r, const Sphere& sp, TriHit& h) const {
        V3 o_err, d_err;
        V3 o = sp.w2o.point_with_error(r.o, o_err);          // transform_ray_with_error (transform.rs:481-510): t_max is NOT shortened here
        const V3 d = sp.w2o.vector_with_error(r.d, d_err);
        const Float l2 = length_squared(d);
        if (l2 > 0.0f) { const Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
        const EFloat ox(o.x, o_err.x), oy(o.y, o_err.y), oz(o.z, o_err.z), dx(d.x, d_err.x), dy(d.y, d_err.y), dz(d.z, d_err.z);
        const EFloat a = dx * dx + dy * dy + dz * dz;
        const EFloat b = EFloat(2.0f) * (dx * ox + dy * oy + dz * oz);
        const EFloat c = ox * ox + oy * oy + oz * oz - EFloat(sp.radius) * EFloat(sp.radius);
        EFloat t0, t1;
        if (!quadratic_efloat(a, b, c, t0, t1)) return false;
        if (t0.upper_bound() > r.t_max || t1.lower_bound() <= 0.0f) return false;
        EFloat t_hit = t0;
        if (t_hit.lower_bound() <= 0.0f) { t_hit = t1; if (t_hit.upper_bound() > r.t_max) return false; }
        V3 p_hit; Float phi;
        auto locate = [&]() {
            p_hit = o + d * t_hit.v;                                          // ray.at(t)
            p_hit = p_hit * (sp.radius / distance(p_hit, V3(0, 0, 0)));      // refine
            if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * sp.radius;
            phi = o_atan2(p_hit.y, p_hit.x);
            if (phi < 0.0f) phi += TWO_PI;
        };
        auto clipped = [&]() { return (sp.z_min > -sp.radius && p_hit.z < sp.z_min) || (sp.z_max < sp.radius && p_hit.z > sp.z_max) || phi > sp.phi_max; };
        locate();
        if (clipped()) {
            if (t_hit.v == t1.v) return false;  // EFloat == compares the values (efloat.rs:143-147)
            if (t1.upper_bound() > r.t_max) return false;
            t_hit = t1;
            locate();
            if (clipped()) return false;
        }
        h.t = t_hit.v; h.b0 = h.b1 = h.b2 = 0.0f; h.sp = p_hit; h.sphi = phi;
        return true;
    }
    void tri_uvs(uint32_t prim, V2 uv[3]) const {  // triangle.rs:384-394
        if (mesh_of(prim).has_uv) { uv[0] = UV[idx[3 * prim]]; uv[1] = UV[idx[3 * prim + 1]]; uv[2] = UV[idx[3 * prim + 2]]; }
        else { uv[0] = V2(0, 0); uv[1] = V2(1, 0); uv[2] = V2(1, 1); }
    }

    // ---- Hyperboloid::intersect / intersect_p up to the accept decision (hyperboloid.rs:124-190 / 266-340).  Fills t, the object-space point, phi, v and the point's error bound
    bool hyperboloid_intersect(const Ray& r, const Hyperboloid& hy, TriHit& h) const {
        V3 o_err, d_err;
        V3 o = hy.w2o.point_with_error(r.o, o_err);
        const V3 d = hy.w2o.vector_with_error(r.d, d_err);
        const Float l2 = length_squared(d);
        if (l2 > 0.0f) { const Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
        const EFloat ox(o.x, o_err.x), oy(o.y, o_err.y), oz(o.z, o_err.z), dx(d.x, d_err.x), dy(d.y, d_err.y), dz(d.z, d_err.z);
        const EFloat ah(hy.ah), ch(hy.ch);
        const EFloat a = ah * dx * dx + ah * dy * dy - ch * dz * dz;
        const EFloat b = EFloat(2.0f) * (ah * dx * ox + ah * dy * oy - ch * dz * oz);
        const EFloat c = ah * ox * ox + ah * oy * oy - ch * oz * oz - EFloat(1.0f);
        EFloat t0, t1;
        if (!quadratic_efloat(a, b, c, t0, t1)) return false;
        if (t0.upper_bound() > r.t_max || t1.lower_bound() <= 0.0f) return false;
        EFloat t_hit = t0;
        if (t_hit.lower_bound() <= 0.0f) { t_hit = t1; if (t_hit.upper_bound() > r.t_max) return false; }
        V3 p_hit; Float phi, v;
        auto locate = [&]() {
            p_hit = o + d * t_hit.v;
            v = (p_hit.z - hy.p1.z) / (hy.p2.z - hy.p1.z);
            const V3 pr = (1.0f - v) * hy.p1 + v * hy.p2;
            phi = o_atan2(pr.x * p_hit.y - p_hit.x * pr.y, p_hit.x * pr.x + p_hit.y * pr.y);
            if (phi < 0.0f) phi += TWO_PI;
        };
        auto clipped = [&]() { return p_hit.z < hy.z_min || p_hit.z > hy.z_max || phi > hy.phi_max; };
        locate();
        if (clipped()) {
            if (t_hit.v == t1.v) return false;
            t_hit = t1;
            if (t1.upper_bound() > r.t_max) return false;
            locate();
            if (clipped()) return false;
        }
        const EFloat px = ox + t_hit * dx, py = oy + t_hit * dy, pz = oz + t_hit * dz;  // :236-243
        h.t = t_hit.v; h.b0 = h.b1 = h.b2 = 0.0f; h.sp = p_hit; h.sphi = phi; h.sv = v;
        h.sperr = V3(px.absolute_error(), py.absolute_error(), pz.absolute_error());
        return true;
    }

    // ---- Cylinder / Cone / Paraboloid / Disk ::intersect and ::intersect_p up to the accept decision (cylinder.rs:64-144, cone.rs:66-130, paraboloid.rs:66-130, disk.rs:64-104)
    bool quadric_intersect(const Ray& r, const Quadric& q, TriHit& h) const {
        V3 o_err, d_err;
        V3 o = q.w2o.point_with_error(r.o, o_err);
        const V3 d = q.w2o.vector_with_error(r.d, d_err);
        const Float l2 = length_squared(d);
        if (l2 > 0.0f) { const Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; }
        if (q.kind == Q_DISK) {
            if (d.z == 0.0f) return false;
            const Float t = (q.height - o.z) / d.z;
            if (t <= 0.0f || t >= r.t_max) return false;
            V3 p = o + d * t;
            const Float dist2 = p.x * p.x + p.y * p.y;
            if (dist2 > q.radius * q.radius || dist2 < q.inner_radius * q.inner_radius) return false;
            Float phi = o_atan2(p.y, p.x);
            if (phi < 0.0f) phi += TWO_PI;
            if (phi > q.phi_max) return false;
            h.t = t; h.b0 = h.b1 = h.b2 = 0.0f; h.sp = p; h.sphi = phi; h.sv = dist2; h.sperr = V3(0, 0, 0);  // sv carries dist2 for the disk
            return true;
        }
        const EFloat ox(o.x, o_err.x), oy(o.y, o_err.y), oz(o.z, o_err.z), dx(d.x, d_err.x), dy(d.y, d_err.y), dz(d.z, d_err.z);
        EFloat a, b, c;
        if (q.kind == Q_CYLINDER) {
            a = dx * dx + dy * dy;
            b = EFloat(2.0f) * (dx * ox + dy * oy);
            c = ox * ox + oy * oy - EFloat(q.radius) * EFloat(q.radius);
        } else if (q.kind == Q_CONE) {
            EFloat k = EFloat(q.radius) / EFloat(q.height);
            k = k * k;
            a = dx * dx + dy * dy - k * dz * dz;
            b = EFloat(2.0f) * (dx * ox + dy * oy - k * dz * (oz - EFloat(q.height)));
            c = ox * ox + oy * oy - k * (oz - EFloat(q.height)) * (oz - EFloat(q.height));
        } else {
            const EFloat k = EFloat(q.z_max) / (EFloat(q.radius) * EFloat(q.radius));
            a = k * (dx * dx + dy * dy);
            b = EFloat(2.0f) * k * (dx * ox + dy * oy) - dz;
            c = k * (ox * ox + oy * oy) - oz;
        }
        EFloat t0, t1;
        if (!quadratic_efloat(a, b, c, t0, t1)) return false;
        if (t0.upper_bound() > r.t_max || t1.lower_bound() <= 0.0f) return false;
        EFloat t_hit = t0;
        if (t_hit.lower_bound() <= 0.0f) { t_hit = t1; if (t_hit.upper_bound() > r.t_max) return false; }
        V3 p_hit; Float phi;
        auto locate = [&]() {
            p_hit = o + d * t_hit.v;
            if (q.kind == Q_CYLINDER) {  // refine (cylinder.rs:107-109)
                const Float hit_rad = std::sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
                p_hit.x *= q.radius / hit_rad; p_hit.y *= q.radius / hit_rad;
            }
            phi = o_atan2(p_hit.y, p_hit.x);
            if (phi < 0.0f) phi += TWO_PI;
        };
        auto clipped = [&]() {
            if (q.kind == Q_CONE) return p_hit.z < 0.0f || p_hit.z > q.height || phi > q.phi_max;
            return p_hit.z < q.z_min || p_hit.z > q.z_max || phi > q.phi_max;
        };
        locate();
        if (clipped()) {
            if (t_hit.v == t1.v) return false;
            t_hit = t1;
            if (t1.upper_bound() > r.t_max) return false;
            locate();
            if (clipped()) return false;
        }
        h.t = t_hit.v; h.b0 = h.b1 = h.b2 = 0.0f; h.sp = p_hit; h.sphi = phi; h.sv = 0.0f;
        if (q.kind == Q_CYLINDER) h.sperr = gamma_n(3) * vabs(V3(p_hit.x, p_hit.y, 0.0f));
        else { const EFloat px = ox + t_hit * dx, py = oy + t_hit * dy, pz = oz + t_hit * dz; h.sperr = V3(px.absolute_error(), py.absolute_error(), pz.absolute_error()); }
        return true;
    }

    // ---- Triangle::intersect up to the accept decision (triangle.rs:438-575 / 731-861).
    // Returns true and fills `h` iff the reference would return Some(..)/true, given constant alpha textures.
    // `test_alpha` mirrors test_alpha_texture; `shadow` selects intersect_p's extra shadow-alpha mask.
    bool tri_intersect(const Ray& r, uint32_t prim, bool test_alpha, bool shadow, TriHit& h) const {
        V3 p0 = P[idx[3 * prim]], p1 = P[idx[3 * prim + 1]], p2 = P[idx[3 * prim + 2]];
        V3 p0t = p0 - r.o, p1t = p1 - r.o, p2t = p2 - r.o;
        int kz = max_dimension(vabs(r.d));
        int kx = kz + 1; if (kx == 3) kx = 0;   // core/src/pbrt/axis.rs:50-55
        int ky = kx + 1; if (ky == 3) ky = 0;
        V3 d = permute(r.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
        Float sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0f / d.z;
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // f64 fallback :483-495
            double a = (double)p2t.x * (double)p1t.y, b = (double)p2t.y * (double)p1t.x;
            e0 = (Float)(b - a);
            a = (double)p0t.x * (double)p2t.y; b = (double)p0t.y * (double)p2t.x;
            e1 = (Float)(b - a);
            a = (double)p1t.x * (double)p0t.y; b = (double)p1t.y * (double)p0t.x;
            e2 = (Float)(b - a);
        }
        if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
        Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < r.t_max * det)) return false;
        else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > r.t_max * det)) return false;
        Float inv_det = 1.0f / det;
        Float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det, t = t_scaled * inv_det;
        Float max_z_t = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
        Float delta_z = gamma_n(3) * max_z_t;
        Float max_x_t = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
        Float max_y_t = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
        Float delta_x = gamma_n(5) * (max_x_t + max_z_t);
        Float delta_y = gamma_n(5) * (max_y_t + max_z_t);
        Float delta_e = 2.0f * (gamma_n(2) * max_x_t * max_y_t + delta_y * max_x_t + delta_x * max_y_t);
        Float max_e = max_component(vabs(V3(e0, e1, e2)));
        Float delta_t = 3.0f * (gamma_n(3) * max_e * max_z_t + delta_e * max_z_t + delta_z * max_e) * std::fabs(inv_det);
        if (t <= delta_t) return false;
        // intersect(): dpdu/dpdv are always computed and a degenerate triangle rejects the hit (:548-574);
        // intersect_p(): the same block runs only under `test_alpha && masks present` (:840-866) — the masks are
        // always Some(ConstantTexture) (:291-312), so it runs whenever test_alpha is true.
        const Mesh& m = mesh_of(prim);
        if (!shadow || test_alpha) {
            if (tri_is_bogus(prim)) return false;
        }
        if (test_alpha) {
            if (m.alpha_tex >= 0 || (shadow && m.shadow_alpha_tex >= 0)) {
                // isect_local: p_hit, uv_hit, no differentials (SurfaceInteraction::new leaves der zero)
                V2 uv[3]; tri_uvs(prim, uv);
                TexCtx c; c.uv = V2((b0 * uv[0].x + b1 * uv[1].x) + b2 * uv[2].x, (b0 * uv[0].y + b1 * uv[1].y) + b2 * uv[2].y);
                c.p = b0 * p0 + b1 * p1 + b2 * p2;
                if (m.alpha_tex >= 0 ? tex_eval(textures, mipmaps, m.alpha_tex, c).c[0] == 0.0f : m.alpha == 0.0f) return false;
                if (shadow && (m.shadow_alpha_tex >= 0 ? tex_eval(textures, mipmaps, m.shadow_alpha_tex, c).c[0] == 0.0f : m.shadow_alpha == 0.0f)) return false;
            } else {
                if (m.alpha == 0.0f) return false;                    // mask.evaluate(..) == 0.0 (:603, :886)
                if (shadow && m.shadow_alpha == 0.0f) return false;   // :891
            }
        }
        h.t = t; h.b0 = b0; h.b1 = b1; h.b2 = b2;
        return true;
    }

    // dpdu/dpdv of triangle.rs:548-574; returns false when "the intersection is bogus" (:567-570)
    bool tri_dpdu_dpdv(uint32_t prim, V3& dpdu, V3& dpdv) const {
        V3 p0 = P[idx[3 * prim]], p1 = P[idx[3 * prim + 1]], p2 = P[idx[3 * prim + 2]];
        V2 uv[3]; tri_uvs(prim, uv);
        V2 duv02(uv[0].x - uv[2].x, uv[0].y - uv[2].y), duv12(uv[1].x - uv[2].x, uv[1].y - uv[2].y);
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = std::fabs(determinant) < 1e-8f;
        dpdu = V3(); dpdv = V3();
        if (!degenerate_uv) {
            Float invdet = 1.0f / determinant;
            dpdu = (duv12.y * dp02 - duv02.y * dp12) * invdet;
            dpdv = (-duv12.x * dp02 + duv02.x * dp12) * invdet;
        }
        if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
            V3 ng = cross(p2 - p0, p1 - p0);
            if (length_squared(ng) == 0.0f) return false;
            coordinate_system(normalize(ng), dpdu, dpdv);
        }
        return true;
    }
    bool tri_is_bogus(uint32_t prim) const { V3 a, b; return !tri_dpdu_dpdv(prim, a, b); }

    // ---- Bounds3::intersect_p_inv (bounds3.rs:292-325), incl. quirk B1 (z far plane not widened)
    static bool box_hit(const Bounds3& b, const Ray& ray, V3 inv_dir, const int neg[3]) {
        Float t_min = (b[neg[0]].x - ray.o.x) * inv_dir.x;
        Float t_max = (b[1 - neg[0]].x - ray.o.x) * inv_dir.x;
        Float t_y_min = (b[neg[1]].y - ray.o.y) * inv_dir.y;
        Float t_y_max = (b[1 - neg[1]].y - ray.o.y) * inv_dir.y;
        Float g3 = gamma_n(3);
        t_max *= 1.0f + 2.0f * g3;
        t_y_max *= 1.0f + 2.0f * g3;
        if (t_min > t_y_max || t_y_min > t_max) return false;
        if (t_y_min > t_min) t_min = t_y_min;
        if (t_y_max < t_max) t_max = t_y_max;
        Float t_z_min = (b[neg[2]].z - ray.o.z) * inv_dir.z;
        Float t_z_max = (b[1 - neg[2]].z - ray.o.z) * inv_dir.z;
        if (t_min > t_z_max || t_z_min > t_max) return false;
        if (t_z_min > t_min) t_min = t_z_min;
        if (t_z_max < t_max) t_max = t_z_max;
        return t_min < ray.t_max && t_max > 0.0f;
    }

    // ---- one primitive of an aggregate: a triangle (GeometricPrimitive::intersect, :67-88, `r.t_max = t`) or a
    //      TransformedPrimitive (transformed_primitive.rs:51-73)
    bool prim_intersect(Ray& r, uint32_t ref, uint32_t& prim_out, TriHit& hit_out, TraversalStats* st) const {
        if (!(ref & ORC_INST_BIT)) {
            TriHit h;
            if (st) st->tri_tests++;
            const Mesh& mm = mesh_of(ref);
            const bool hit = mm.sphere >= 0 ? sphere_intersect(r, spheres[(size_t)mm.sphere], h) : (mm.hyper >= 0 ? hyperboloid_intersect(r, hyperboloids[(size_t)mm.hyper], h) : (mm.quadric >= 0 ? quadric_intersect(r, quadrics[(size_t)mm.quadric], h) : tri_intersect(r, ref, true, false, h)));
            if (hit) { r.t_max = h.t; prim_out = ref; hit_out = h; return true; }
            return false;
        }
        const uint32_t ii = ref & ~ORC_INST_BIT;
        const Instance& in = instances[ii]; const Object& ob = objects[in.object];
        Ray ray = transform_ray(in.i2w.inv(), r);
        bool hit;
        if (ob.tri1 - ob.tri0 == 1) hit = prim_intersect(ray, ob.tri0, prim_out, hit_out, st);
        else hit = intersect_in(ob.nodes, ob.ordered_prims, ray, prim_out, hit_out, st);
        if (!hit) return false;
        r.t_max = ray.t_max;
        hit_out.inst = ii + 1;
        return true;
    }
    bool prim_intersect_p(const Ray& r, uint32_t ref, TraversalStats* st) const {
        if (!(ref & ORC_INST_BIT)) {
            TriHit h; if (st) st->tri_tests++;
            if (mesh_of(ref).sphere >= 0) return sphere_intersect(r, spheres[(size_t)mesh_of(ref).sphere], h);
            if (mesh_of(ref).hyper >= 0) return hyperboloid_intersect(r, hyperboloids[(size_t)mesh_of(ref).hyper], h);
            if (mesh_of(ref).quadric >= 0) return quadric_intersect(r, quadrics[(size_t)mesh_of(ref).quadric], h);
            return tri_intersect(r, ref, true, true, h);
        }
        const Instance& in = instances[ref & ~ORC_INST_BIT]; const Object& ob = objects[in.object];
        Ray ray = transform_ray(in.i2w.inv(), r);
        if (ob.tri1 - ob.tri0 == 1) return prim_intersect_p(ray, ob.tri0, st);
        return intersect_p_in(ob.nodes, ob.ordered_prims, ray, st);
    }

    // ---- BVHAccel::intersect (bvh/mod.rs:173-226)
    bool intersect_in(const std::vector<LinearBVHNode>& nodes, const std::vector<uint32_t>& ordered_prims, Ray& r, uint32_t& prim_out, TriHit& hit_out,
                      TraversalStats* st) const {
        bool any = false;
        if (nodes.empty()) return false;
        V3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        int neg[3] = {inv_dir.x < 0.0f ? 1 : 0, inv_dir.y < 0.0f ? 1 : 0, inv_dir.z < 0.0f ? 1 : 0};
        size_t to_visit = 0, cur = 0;
        size_t stack[64];
        for (;;) {
            const LinearBVHNode& node = nodes[cur];
            if (st) st->nodes_visited++;
            if (box_hit(node.bounds, r, inv_dir, neg)) {
                if (node.n_primitives > 0) {
                    for (uint32_t i = 0; i < node.n_primitives; i++)
                        if (prim_intersect(r, ordered_prims[node.offset + i], prim_out, hit_out, st)) any = true;
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[node.axis] == 1) { stack[to_visit++] = cur + 1; cur = node.offset; }
                    else { stack[to_visit++] = node.offset; cur = cur + 1; }
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return any;
    }
    bool intersect(Ray& r, uint32_t& prim_out, TriHit& hit_out, TraversalStats* st = nullptr) const {
        if (st) st->rays++;
        hit_out.inst = 0;
        return intersect_in(nodes, ordered_prims, r, prim_out, hit_out, st);
    }
    // ---- BVHAccel::intersect_p (bvh/mod.rs:231-283)
    bool intersect_p_in(const std::vector<LinearBVHNode>& nodes, const std::vector<uint32_t>& ordered_prims, const Ray& r, TraversalStats* st) const {
        if (nodes.empty()) return false;
        V3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        int neg[3] = {inv_dir.x < 0.0f ? 1 : 0, inv_dir.y < 0.0f ? 1 : 0, inv_dir.z < 0.0f ? 1 : 0};
        size_t to_visit = 0, cur = 0;
        size_t stack[64];
        for (;;) {
            const LinearBVHNode& node = nodes[cur];
            if (st) st->nodes_visited++;
            if (box_hit(node.bounds, r, inv_dir, neg)) {
                if (node.n_primitives > 0) {
                    for (uint32_t i = 0; i < node.n_primitives; i++)
                        if (prim_intersect_p(r, ordered_prims[node.offset + i], st)) return true;
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[node.axis] == 1) { stack[to_visit++] = cur + 1; cur = node.offset; }
                    else { stack[to_visit++] = node.offset; cur = cur + 1; }
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return false;
    }
    bool intersect_p(const Ray& r, TraversalStats* st = nullptr) const {
        if (st) st->rays++;
        return intersect_p_in(nodes, ordered_prims, r, st);
    }

    // ================= BVH build: BVHAccel::new + sah::build (bvh/mod.rs:43-153, sah.rs:26-367) ================
    struct PrimInfo { size_t number; Bounds3 bounds; V3 centroid; };

    // itertools::partition (published algorithm, itertools 0.13 src/lib.rs `pub fn partition`)
    template <class Pred> static size_t itertools_partition(PrimInfo* first, PrimInfo* last, Pred pred) {
        size_t split = 0;
        PrimInfo* front = first;
        PrimInfo* back = last;
        while (front != back) {
            PrimInfo* f = front++;
            if (!pred(*f)) {
                bool found = false;
                while (front != back) {
                    PrimInfo* b = --back;
                    if (pred(*b)) { std::swap(*f, *b); found = true; break; }
                }
                if (!found) return split;
            }
            split++;
        }
        return split;
    }

    static size_t sah_bucket(const Bounds3& cb, V3 c, int dim) {  // sah.rs:309-313
        size_t b = f2usize(12.0f * cb.offset(c)[dim]);
        if (b == 12) b = 11;
        return b;
    }

    // returns node index; split_method: 0 SAH, 2 middle (quirk B6), 3 equal counts
    uint32_t build_rec(std::vector<PrimInfo>& info, size_t start, size_t end, int split_method) {
        uint32_t my = (uint32_t)nodes.size();
        nodes.emplace_back();
        Bounds3 bounds;
        for (size_t i = start; i < end; i++) bounds = bounds.union_b(info[i].bounds);
        size_t n = end - start;
        auto make_leaf = [&]() {
            uint32_t first = (uint32_t)ordered_prims.size();
            for (size_t i = start; i < end; i++) ordered_prims.push_back((uint32_t)info[i].number);
            nodes[my].bounds = bounds; nodes[my].offset = first; nodes[my].n_primitives = (uint16_t)n; nodes[my].axis = 0; nodes[my].pad = 0;
            return my;
        };
        if (n == 1) return make_leaf();
        Bounds3 cb;
        for (size_t i = start; i < end; i++) cb = cb.union_p(info[i].centroid);
        int dim = cb.maximum_extent();
        if (cb.pmax[dim] == cb.pmin[dim]) return make_leaf();
        size_t mid = 0; bool have_mid = true;
        auto equal_counts = [&]() {
            size_t m = (start + end) / 2;
            std::nth_element(info.begin() + start, info.begin() + m, info.begin() + end,
                             [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
            return m;
        };
        if (split_method == 3) mid = equal_counts();
        else if (split_method == 2) {  // sah.rs:61-76 incl. quirk B6
            Float pmid = (cb.pmin[dim] + cb.pmax[dim]) / 2.0f;
            size_t m = start + itertools_partition(&info[start], &info[start] + n, [&](const PrimInfo& pi) { return pi.centroid[dim] < pmid; });
            if (m != start && m != end) m = equal_counts();
            mid = m;
        } else if (n <= 2) {
            mid = (start + end) / 2;  // split_equal_counts on 2 elements: [smaller, larger]
            if (info[start + 1].centroid[dim] < info[start].centroid[dim]) std::swap(info[start], info[start + 1]);
        } else {
            // split_sah (sah.rs:293-367)
            size_t count[12] = {0}; Bounds3 bb[12];
            for (size_t i = start; i < end; i++) {
                size_t b = sah_bucket(cb, info[i].centroid, dim);
                count[b]++; bb[b] = bb[b].union_b(info[i].bounds);
            }
            Float cost[11];
            for (int i = 0; i < 11; i++) {
                Bounds3 b0, b1; size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { b0 = b0.union_b(bb[j]); c0 += count[j]; }
                for (int j = i + 1; j < 12; j++) { b1 = b1.union_b(bb[j]); c1 += count[j]; }
                cost[i] = 1.0f + ((Float)c0 * b0.surface_area() + (Float)c1 * b1.surface_area()) / bounds.surface_area();
            }
            Float min_cost = cost[0]; size_t min_b = 0;
            for (size_t i = 1; i < 11; i++) if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
            Float leaf_cost = (Float)n;
            if (n > (size_t)max_prims_in_node || min_cost < leaf_cost) {
                mid = start + itertools_partition(&info[start], &info[start] + n,
                                                  [&](const PrimInfo& pi) { return sah_bucket(cb, pi.centroid, dim) <= min_b; });
            } else have_mid = false;
        }
        if (!have_mid) return make_leaf();
        if (mid == start || mid == end) {
            // the reference recurses with start==end and hits assert_ne! (sah.rs:37): a panic.  The oracle reports it.
            std::fprintf(stderr, "oracle: BVH split produced an empty side (reference would panic, sah.rs:37)\n");
            return make_leaf();
        }
        build_rec(info, start, mid, split_method);
        uint32_t c1 = build_rec(info, mid, end, split_method);
        // interior bounds = c0.bounds ∪ c1.bounds (common.rs:152-160); c0 sits at my+1 (mod.rs:126-153)
        nodes[my].bounds = nodes[my + 1].bounds.union_b(nodes[c1].bounds);
        nodes[my].offset = c1; nodes[my].n_primitives = 0; nodes[my].axis = (uint8_t)dim; nodes[my].pad = 0;
        return my;
    }

    // ================= HLBVH: hlbvh::build (bvh/hlbvh.rs:33-449) + morton.rs ====================================
    // Restated as written, including quirk B10: encode_morton_3 feeds left_shift_3 the IEEE BIT PATTERN of the scaled
    // centroid offset (`float_to_bits`, morton.rs:33-39), not its integer value, so the "Morton" order is an order on
    // mantissa bits.  The tree is valid (every primitive lands in exactly one leaf) but not spatially coherent.
    // ordered_prims offsets are handed out in the single-thread order (treelet by treelet, depth first); with several
    // threads the reference assigns them in completion order, which changes nothing a ray can observe.
    struct MortonPrim { size_t primitive_index; uint32_t morton_code; };
    struct HNode { Bounds3 bounds; int kid[2]; uint32_t first, n; int axis; };
    std::vector<HNode> hpool;
    bool hlbvh_panic = false;

    static uint32_t left_shift_3(uint32_t x) {  // morton.rs:101-118 (wrapping shifts on u32)
        uint32_t x1 = (x == (1u << 10)) ? x - 1 : x;
        x1 = (x1 | (x1 << 16)) & 0x030000FFu;
        x1 = (x1 | (x1 << 8)) & 0x0300F00Fu;
        x1 = (x1 | (x1 << 4)) & 0x030C30C3u;
        x1 = (x1 | (x1 << 2)) & 0x09249249u;
        return x1;
    }
    static uint32_t encode_morton_3(V3 v) {
        return (left_shift_3(float_to_bits(v.z)) << 2) | (left_shift_3(float_to_bits(v.y)) << 1) | left_shift_3(float_to_bits(v.x));
    }
    static void radix_sort(std::vector<MortonPrim>& v) {  // morton.rs:50-98: 5 stable passes of 6 bits
        std::vector<MortonPrim> tmp(v.size());
        for (int pass = 0; pass < 5; pass++) {
            int low_bit = pass * 6;
            std::vector<MortonPrim>& in = (pass & 1) ? tmp : v;
            std::vector<MortonPrim>& out = (pass & 1) ? v : tmp;
            size_t count[64] = {0}, out_index[64];
            for (const MortonPrim& mp : in) count[(mp.morton_code >> low_bit) & 63]++;
            out_index[0] = 0;
            for (int i = 1; i < 64; i++) out_index[i] = out_index[i - 1] + count[i - 1];
            for (const MortonPrim& mp : in) out[out_index[(mp.morton_code >> low_bit) & 63]++] = mp;
        }
        v.swap(tmp);  // N_PASSES is odd: the result sits in the temporary
    }
    int emit_lbvh(const std::vector<PrimInfo>& info, const MortonPrim* mp, size_t n, int bit_index) {  // hlbvh.rs:199-294
        if (bit_index == -1 || n < (size_t)max_prims_in_node) {
            HNode h; h.kid[0] = h.kid[1] = -1; h.axis = 0; h.n = (uint32_t)n;
            h.first = (uint32_t)ordered_prims.size();
            for (size_t i = 0; i < n; i++) {
                ordered_prims.push_back((uint32_t)info[mp[i].primitive_index].number);
                h.bounds = h.bounds.union_b(info[mp[i].primitive_index].bounds);
            }
            hpool.push_back(h);
            return (int)hpool.size() - 1;
        }
        uint32_t mask = 1u << bit_index;
        if ((mp[0].morton_code & mask) == (mp[n - 1].morton_code & mask)) return emit_lbvh(info, mp, n, bit_index - 1);
        size_t search_start = 0, search_end = n - 1;
        while (search_start + 1 != search_end) {
            size_t mid = (search_start + search_end) / 2;
            if ((mp[search_start].morton_code & mask) == (mp[mid].morton_code & mask)) search_start = mid;
            else search_end = mid;
        }
        size_t split = search_end;
        int c0 = emit_lbvh(info, mp, split, bit_index - 1);
        int c1 = emit_lbvh(info, mp + split, n - split, bit_index - 1);
        HNode h; h.kid[0] = c0; h.kid[1] = c1; h.first = 0; h.n = 0; h.axis = bit_index % 3;
        h.bounds = hpool[c0].bounds.union_b(hpool[c1].bounds);
        hpool.push_back(h);
        return (int)hpool.size() - 1;
    }
    static size_t hl_bucket(Float centroid, Float lo, Float hi) {  // hlbvh.rs:349-355
        size_t b = f2usize(12.0f * ((centroid - lo) / (hi - lo)));
        if (b == 12) b = 11;
        return b;
    }
    int build_upper_sah(std::vector<int>& roots, size_t start, size_t end) {  // hlbvh.rs:296-432
        size_t n_nodes = end - start;
        if (n_nodes == 1) return roots[start];
        Bounds3 bounds, cb;
        for (size_t i = start; i < end; i++) bounds = bounds.union_b(hpool[roots[i]].bounds);
        for (size_t i = start; i < end; i++) cb = cb.union_p((hpool[roots[i]].bounds.pmin + hpool[roots[i]].bounds.pmax) * 0.5f);
        int dim = cb.maximum_extent();
        if (cb.pmax[dim] == cb.pmin[dim]) { hlbvh_panic = true; return roots[start]; }  // assert_ne! (:338)
        size_t count[12] = {0}; Bounds3 bb[12];
        for (size_t i = start; i < end; i++) {
            const Bounds3& nb = hpool[roots[i]].bounds;
            Float c = (nb.pmin[dim] + nb.pmax[dim]) * 0.5f;
            size_t b = hl_bucket(c, cb.pmin[dim], cb.pmax[dim]);
            if (b >= 12) { hlbvh_panic = true; return roots[start]; }  // assert!(b < N_BUCKETS)
            count[b]++; bb[b] = bb[b].union_b(nb);
        }
        Float cost[11];
        for (int i = 0; i < 11; i++) {
            Bounds3 b0, b1; size_t c0 = 0, c1 = 0;
            for (int j = 0; j <= i; j++) { b0 = b0.union_b(bb[j]); c0 += count[j]; }
            for (int j = i + 1; j < 12; j++) { b1 = b1.union_b(bb[j]); c1 += count[j]; }
            cost[i] = 0.125f + ((Float)c0 * b0.surface_area() + (Float)c1 * b1.surface_area()) / bounds.surface_area();
        }
        Float min_cost = cost[0]; size_t min_b = 0;
        for (size_t i = 1; i < 11; i++) if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
        // itertools::partition over the treelet roots
        size_t split = 0;
        {
            size_t front = start, back = end;
            auto pred = [&](int r) {
                const Bounds3& nb = hpool[r].bounds;
                Float c = (nb.pmin[dim] + nb.pmax[dim]) * 0.5f;
                return hl_bucket(c, cb.pmin[dim], cb.pmax[dim]) <= min_b;
            };
            while (front != back) {
                size_t f = front++;
                if (!pred(roots[f])) {
                    bool found = false;
                    while (front != back) { size_t b = --back; if (pred(roots[b])) { std::swap(roots[f], roots[b]); found = true; break; } }
                    if (!found) break;
                }
                split++;
            }
        }
        size_t mid = start + split;
        if (!(mid > start) || !(mid < end)) { hlbvh_panic = true; return roots[start]; }  // assert! (:418-419)
        int c0 = build_upper_sah(roots, start, mid);
        int c1 = build_upper_sah(roots, mid, end);
        HNode h; h.kid[0] = c0; h.kid[1] = c1; h.first = 0; h.n = 0; h.axis = dim;
        h.bounds = hpool[c0].bounds.union_b(hpool[c1].bounds);
        hpool.push_back(h);
        return (int)hpool.size() - 1;
    }
    uint32_t flatten_h(int node) {  // flatten_bvh_tree (bvh/mod.rs:118-150)
        uint32_t my = (uint32_t)nodes.size();
        nodes.emplace_back();
        const HNode h = hpool[node];
        if (h.n > 0) {
            nodes[my].bounds = h.bounds; nodes[my].offset = h.first; nodes[my].n_primitives = (uint16_t)h.n; nodes[my].axis = 0; nodes[my].pad = 0;
            if (h.n >= 65536) hlbvh_panic = true;  // assert!(node.n_primitives < 65536)
        } else {
            flatten_h(h.kid[0]);
            uint32_t c1 = flatten_h(h.kid[1]);
            nodes[my].bounds = h.bounds; nodes[my].offset = c1; nodes[my].n_primitives = 0; nodes[my].axis = (uint8_t)h.axis; nodes[my].pad = 0;
        }
        return my;
    }
    void build_hlbvh(std::vector<PrimInfo>& info) {
        Bounds3 bounds;
        for (const PrimInfo& pi : info) bounds = bounds.union_b(pi.bounds);
        std::vector<MortonPrim> mp(info.size());
        for (size_t i = 0; i < info.size(); i++) {
            V3 v = bounds.offset(info[i].centroid) * 1024.0f;  // MORTON_SCALE = 1 << 10
            mp[i].primitive_index = i; mp[i].morton_code = encode_morton_3(v);  // primitive_number == position in the aggregate's list
        }
        radix_sort(mp);
        hpool.clear(); hlbvh_panic = false;
        std::vector<int> roots;
        const uint32_t MASK = 0x3FFC0000u;
        size_t start = 0;
        for (size_t end = 1; end <= mp.size(); end++) {
            if (end == mp.size() || ((mp[start].morton_code & MASK) != (mp[end].morton_code & MASK))) {
                roots.push_back(emit_lbvh(info, &mp[start], end - start, 29 - 12));  // FIRST_BIT_INDEX = N_BITS - 1 - N_BUCKETS
                start = end;
            }
        }
        int root = build_upper_sah(roots, 0, roots.size());
        flatten_h(root);
    }

    // builds over `info` into the members nodes / ordered_prims
    void build_over(std::vector<PrimInfo>& info, int split_method) {
        nodes.clear(); ordered_prims.clear();
        if (info.empty()) return;
        nodes.reserve(2 * info.size()); ordered_prims.reserve(info.size());
        if (split_method == 1) build_hlbvh(info);
        else build_rec(info, 0, info.size(), split_method);
    }
    PrimInfo info_of(size_t number, const Bounds3& b) const {
        PrimInfo pi; pi.number = number; pi.bounds = b; pi.centroid = 0.5f * (b.pmin + b.pmax);  // common.rs:85-91
        return pi;
    }
    void build_bvh(int split_method, int max_prims) {
        max_prims_in_node = max_prims & 0xff;  // quirk B8: `as u8`
        // aggregates of the instanced objects first (make_accelerator at ObjectInstance time, lib.rs:953-971)
        for (const Instance& in : instances) {
            Object& ob = objects[in.object];
            if (ob.built) continue;
            ob.built = true;
            const uint32_t n = ob.tri1 - ob.tri0;
            if (n == 1) { ob.bound = tri_bound(ob.tri0); continue; }
            std::vector<PrimInfo> info;
            for (uint32_t t = ob.tri0; t < ob.tri1; t++) info.push_back(info_of(t, tri_bound(t)));
            build_over(info, split_method);
            ob.nodes.swap(nodes); ob.ordered_prims.swap(ordered_prims);
            ob.bound = ob.nodes.empty() ? Bounds3() : ob.nodes[0].bounds;  // BVHAccel::world_bound (bvh/mod.rs:161-167)
        }
        // the scene aggregate over triangles and TransformedPrimitives (world_bound = motion_bounds = transform_bounds when static)
        std::vector<PrimInfo> info;
        if (objects.empty()) for (size_t i = 0; i < n_tris(); i++) info.push_back(info_of(i, tri_bound((uint32_t)i)));
        else for (uint32_t it : top_items) {
            if (it & ORC_INST_BIT) { const Instance& in = instances[it & ~ORC_INST_BIT]; info.push_back(info_of(it, transform_bounds(in.i2w, objects[in.object].bound))); }
            else info.push_back(info_of(it, tri_bound(it)));
        }
        build_over(info, split_method);
        if (nodes.empty()) return;
        world_bound = nodes[0].bounds;
        world_bound.bounding_sphere(world_center, world_radius);  // Light::preprocess (infinite.rs:113-117, distant.rs:54-58)
    }
};

}  // namespace orc
