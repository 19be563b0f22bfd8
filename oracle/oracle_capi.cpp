// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// C API over the CPU restatement, mirroring include/pbrt_hip.h one-for-one with the prefix `oracle_` so the same
// Python scene-description code can drive both the checker (this) and the product (libpbrt_hip.so).
#include "oracle_render.hpp"
#include <chrono>
#include <cstdlib>

using namespace orc;

struct OracleScene {
    Scene sc;
    Renderer r;
    std::string err;
    bool built = false, have_camera = false, have_film = false, have_sampler = false;
    std::vector<uint32_t> sobol32; std::vector<uint64_t> vdc, vdc_inv;
    RayRecorder rec;
    OracleScene() { r.sc = &sc; }
};

struct OracleRay { float o[3]; float t_max; float d[3]; float time; };
struct OracleHit { float t; uint32_t prim; float b0, b1, b2; uint32_t pad[3]; };
struct OracleStats {
    uint64_t camera_rays, regular_rays, shadow_rays, paths_zero_radiance, paths_total;
    double render_seconds, extend_seconds, shadow_seconds, shade_seconds;
    uint64_t extend_launches, shadow_launches, light_distributions_created;
};
struct OracleTraversalStats { uint64_t rays, nodes_visited, tri_tests; };

static M4 m4_from(const float* a) { M4 m; std::memcpy(m.m, a, 64); return m; }

extern "C" {

int oracle_device_count(void) { return 0; }
OracleScene* oracle_scene_create(int) { return new OracleScene(); }
void oracle_scene_destroy(OracleScene* s) { delete s; }
const char* oracle_last_error(const OracleScene* s) { return s ? s->err.c_str() : "null handle"; }

// ---- materials: the BxDF list compute_scattering_functions builds, for constant textures (materials/src/*.rs) -------------
static Float roughness_to_alpha(Float roughness) {  // microfacet/trowbridge_reitz.rs:31-40
    roughness = pmax(roughness, 1e-3f);
    Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static void set_tr(Lobe& l, Float ax, Float ay) { l.ax = pmax(0.001f, ax); l.ay = pmax(0.001f, ay); }  // TrowbridgeReitzDistribution::new (:21-28)
static Spec spec3(const float v[3]) { return Spec(v[0], v[1], v[2]); }
static int push_material(OracleScene* s, Material& m, uint32_t* out_id) {
    s->sc.materials.push_back(m);
    if (out_id) *out_id = (uint32_t)s->sc.materials.size() - 1;
    return 0;
}
int oracle_add_material_matte(OracleScene* s, const float kd[3], float sigma, uint32_t* out_id) {  // matte.rs:47-76
    if (!s || !kd) return -1;
    Material m; m.kd = spec3(kd); m.sigma = sigma;
    Spec r = spec_clamp0(m.kd);
    Float sig = pclamp(sigma, 0.0f, 90.0f);
    if (!r.is_black()) {
        Lobe l; l.r = r; l.type = BX_REFL | BX_DIFF;
        if (sig == 0.0f) l.kind = LK_LAMBERT;
        else {  // oren_nayar.rs:28-39
            l.kind = LK_OREN;
            Float sg = to_radians(sig), s2 = sg * sg;
            l.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
            l.b = 0.45f * s2 / (s2 + 0.09f);
        }
        m.lobes.push_back(l);
        m.param_lobe[0] = 0;
    }
    return push_material(s, m, out_id);
}
int oracle_add_material_none(OracleScene* s, uint32_t* out_id) {  // Material "none" / "": GeometricPrimitive without a material (geometric_primitive.rs:112-134)
    if (!s) return -1;
    Material m; m.none = true;
    return push_material(s, m, out_id);
}
int oracle_add_material_mirror(OracleScene* s, const float kr[3], uint32_t* out_id) {  // mirror.rs:40-62
    if (!s || !kr) return -1;
    Material m; m.general = true;
    Spec r = spec_clamp0(spec3(kr));
    if (!r.is_black()) { Lobe l; l.kind = LK_SPEC_R; l.type = BX_REFL | BX_SPEC; l.fresnel = FR_NOOP; l.r = r; m.lobes.push_back(l); m.param_lobe[2] = 0; }
    return push_material(s, m, out_id);
}
int oracle_add_material_plastic(OracleScene* s, const float kd[3], const float ks[3], float roughness, int remap, uint32_t* out_id) {  // plastic.rs:50-82
    if (!s || !kd || !ks) return -1;
    Material m; m.general = true;
    Spec d = spec_clamp0(spec3(kd)), sp = spec_clamp0(spec3(ks));
    if (!d.is_black()) { Lobe l; l.kind = LK_LAMBERT; l.type = BX_REFL | BX_DIFF; l.r = d; m.lobes.push_back(l); m.param_lobe[0] = 0; }
    if (!sp.is_black()) {
        m.param_lobe[1] = (int)m.lobes.size();
        Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.5f; l.eta_b = 1.0f; l.r = sp;
        Float rough = remap ? roughness_to_alpha(roughness) : roughness;
        set_tr(l, rough, rough);
        m.rough_lobe = (int)m.lobes.size(); m.rough_remap = remap != 0;
        m.lobes.push_back(l);
    }
    return push_material(s, m, out_id);
}
int oracle_add_material_glass(OracleScene* s, const float kr[3], const float kt[3], float urough, float vrough, float eta, int remap, uint32_t* out_id) {  // glass.rs:62-118
    if (!s || !kr || !kt) return -1;
    Material m; m.general = true;  // BSDF::new(.., None): eta stays 1 (glass.rs:82)
    m.made_as = 2; m.raw_k[2] = spec3(kr); m.raw_k[3] = spec3(kt); m.raw_eta = eta; m.raw_ur = urough; m.raw_vr = vrough; m.raw_remap = remap != 0;
    Spec r = spec_clamp0(spec3(kr)), t = spec_clamp0(spec3(kt));
    if (!(r.is_black() && t.is_black())) {
        bool is_specular = urough == 0.0f && vrough == 0.0f;
        if (is_specular) {  // allow_multiple_lobes is true on this path (path.rs:143)
            Lobe l; l.kind = LK_FRESNEL_SPEC; l.type = BX_REFL | BX_TRANS | BX_SPEC; l.r = r; l.t = t; l.eta_a = 1.0f; l.eta_b = eta; m.lobes.push_back(l);
            m.param_lobe[2] = 0; m.param_field[2] = 0; m.param_lobe[3] = 0; m.param_field[3] = 1;
        } else {
            if (remap) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
            if (!r.is_black()) { Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; l.r = r; set_tr(l, urough, vrough); m.param_lobe[2] = (int)m.lobes.size(); m.lobes.push_back(l); }
            if (!t.is_black()) { m.param_lobe[3] = (int)m.lobes.size(); m.param_field[3] = 1; Lobe l; l.kind = LK_MICRO_T; l.type = BX_TRANS | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; l.t = t; set_tr(l, urough, vrough); m.lobes.push_back(l); }
        }
    }
    return push_material(s, m, out_id);
}
int oracle_add_material_metal(OracleScene* s, const float eta[3], const float k[3], float urough, float vrough, int remap, uint32_t* out_id) {  // metal.rs:62-98
    if (!s || !eta || !k) return -1;
    Material m; m.general = true; m.made_as = 3;
    if (remap) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
    Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_COND; l.r = Spec(1.0f);
    l.c_eta_i = Spec(1.0f); l.c_eta_t = spec3(eta); l.c_k = spec3(k);
    set_tr(l, urough, vrough);
    m.rough_lobe = 0; m.rough_remap = remap != 0;
    m.lobes.push_back(l);
    return push_material(s, m, out_id);
}
int oracle_add_material_uber(OracleScene* s, const float kd[3], const float ks[3], const float kr[3], const float kt[3], const float opacity[3], float urough,
                             float vrough, float eta, int remap, uint32_t* out_id) {  // uber.rs:116-186
    if (!s || !kd || !ks || !kr || !kt || !opacity) return -1;
    Material m; m.general = true;
    m.made_as = 1; m.raw_k[0] = spec3(kd); m.raw_k[1] = spec3(ks); m.raw_k[2] = spec3(kr); m.raw_k[3] = spec3(kt); m.raw_eta = eta; m.raw_ur = urough; m.raw_vr = vrough; m.raw_remap = remap != 0;
    Float e = eta;
    Spec op = spec_clamp0(spec3(opacity));
    Spec t = spec_clamp0(-op + Spec(1.0f));
    if (!t.is_black()) {
        m.bsdf_eta = 1.0f;
        Lobe l; l.kind = LK_SPEC_T; l.type = BX_TRANS | BX_SPEC; l.fresnel = FR_DIEL; l.t = t; l.eta_a = 1.0f; l.eta_b = 1.0f; m.lobes.push_back(l);
    } else m.bsdf_eta = e;
    Spec d = op * spec_clamp0(spec3(kd));
    m.has_pre = true; m.pre = op;
    if (!d.is_black()) { Lobe l; l.kind = LK_LAMBERT; l.type = BX_REFL | BX_DIFF; l.r = d; m.param_lobe[0] = (int)m.lobes.size(); m.lobes.push_back(l); }
    Spec sp = op * spec_clamp0(spec3(ks));
    if (!sp.is_black()) {
        m.param_lobe[1] = (int)m.lobes.size();
        Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r = sp;
        if (remap) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
        set_tr(l, urough, vrough);
        m.rough_lobe = (int)m.lobes.size(); m.rough_remap = remap != 0;
        m.lobes.push_back(l);
    }
    Spec r = op * spec_clamp0(spec3(kr));
    if (!r.is_black()) { Lobe l; l.kind = LK_SPEC_R; l.type = BX_REFL | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r = r; m.param_lobe[2] = (int)m.lobes.size(); m.lobes.push_back(l); }
    Spec tt = op * spec_clamp0(spec3(kt));
    if (!tt.is_black()) { m.param_lobe[3] = (int)m.lobes.size(); m.param_field[3] = 1; Lobe l; l.kind = LK_SPEC_T; l.type = BX_TRANS | BX_SPEC; l.fresnel = FR_DIEL; l.t = tt; l.eta_a = 1.0f; l.eta_b = e; m.lobes.push_back(l); }
    return push_material(s, m, out_id);
}

int oracle_add_mesh(OracleScene* s, const float* P, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, const float* N,
                    const float* S, const float* UV, uint32_t material_id, int32_t first_area_light_id, uint32_t flags,
                    float alpha, float shadow_alpha) {
    if (!s || !P || !indices) return -1;
    if (material_id >= s->sc.materials.size()) { s->err = "bad material id"; return -1; }
    for (uint32_t i = 0; i < 3 * n_tris; i++) if (indices[i] >= n_verts) { s->err = "vertex index out of bounds"; return -1; }
    Scene& sc = s->sc;
    if (sc.open_object >= 0 && first_area_light_id >= 0) {
        // "Area lights not supported with object instancing" (api/src/lib.rs:877-881): the primitives keep their area light (its emission), the scene's light list does not get it
        if ((size_t)first_area_light_id + n_tris != sc.lights.size()) { s->err = "area lights of a shape inside an object definition must be the ones created last"; return -1; }
        for (uint32_t k = 0; k < n_tris; k++) if (sc.lights[first_area_light_id + k].type != L_AREA) { s->err = "light is not an area light"; return -1; }
        const int32_t first = -2 - (int32_t)sc.emission_only.size();
        sc.emission_only.insert(sc.emission_only.end(), sc.lights.begin() + first_area_light_id, sc.lights.end());
        sc.lights.resize((size_t)first_area_light_id);
        first_area_light_id = first;
        s->err = "warning: Area lights not supported with object instancing.";
    }
    Mesh m;
    m.vert_base = (uint32_t)sc.P.size(); m.tri_base = (uint32_t)sc.n_tris(); m.n_verts = n_verts; m.n_tris = n_tris;
    m.has_n = N != nullptr; m.has_s = S != nullptr; m.has_uv = UV != nullptr;
    m.material = material_id; m.first_light = first_area_light_id;
    m.reverse_orientation = flags & 1; m.swaps_handedness = (flags >> 1) & 1;
    m.alpha = alpha; m.shadow_alpha = shadow_alpha;
    // P/N/S/UV arrays are kept index-aligned with P (zero-filled where a mesh lacks them)
    for (uint32_t i = 0; i < n_verts; i++) {
        sc.P.push_back(V3(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
        sc.N.push_back(N ? V3(N[3 * i], N[3 * i + 1], N[3 * i + 2]) : V3());
        sc.S.push_back(S ? V3(S[3 * i], S[3 * i + 1], S[3 * i + 2]) : V3());
        sc.UV.push_back(UV ? V2(UV[2 * i], UV[2 * i + 1]) : V2());
    }
    uint32_t mesh_id = (uint32_t)sc.meshes.size();
    for (uint32_t i = 0; i < 3 * n_tris; i++) sc.idx.push_back(indices[i] + m.vert_base);
    for (uint32_t i = 0; i < n_tris; i++) sc.tri_mesh.push_back(mesh_id);
    sc.meshes.push_back(m);
    if (first_area_light_id >= 0) {
        if ((size_t)first_area_light_id + n_tris > sc.lights.size()) { s->err = "area light ids out of range"; return -1; }
        for (uint32_t k = 0; k < n_tris; k++) {
            Light& l = sc.lights[first_area_light_id + k];
            if (l.type != L_AREA) { s->err = "light is not an area light"; return -1; }
            uint32_t prim = m.tri_base + k;
            l.prim = prim;
            V3 p0 = sc.P[sc.idx[3 * prim]], p1 = sc.P[sc.idx[3 * prim + 1]], p2 = sc.P[sc.idx[3 * prim + 2]];
            l.area = 0.5f * length(cross(p1 - p0, p2 - p0));  // Triangle::area (triangle.rs:906-911)
        }
    }
    if (sc.open_object >= 0) sc.objects[sc.open_object].tri1 = m.tri_base + n_tris;
    else for (uint32_t k = 0; k < n_tris; k++) sc.top_items.push_back(m.tri_base + k);
    s->built = false;
    return 0;
}

// Shape "sphere" (shapes/src/sphere.rs:494-520 reads radius / zmin / zmax / phimax) — ORACLE ONLY, for BASELINE.json's configs[0].  object_to_world is handed over
// as the reference's Transform holds it: the matrix AND the inverse the CTM accumulated (transform.rs:644-656 multiplies the inverses, it never inverts a product).
// The sphere takes one slot of the scene's primitive list, in directive order with the meshes.
int oracle_add_sphere(OracleScene* s, const float o2w_m[16], const float o2w_minv[16], float radius, float z_min, float z_max, float phi_max, uint32_t material_id, uint32_t flags) {
    if (!s || !o2w_m || !o2w_minv) return -1;
    Scene& sc = s->sc;
    if (material_id >= sc.materials.size()) { s->err = "bad material id"; return -1; }
    if (sc.open_object >= 0) { s->err = "spheres inside object instances are not restated"; return -5; }
    const Transform o2w(m4_from(o2w_m), m4_from(o2w_minv));
    Mesh m;
    m.vert_base = (uint32_t)sc.P.size(); m.tri_base = (uint32_t)sc.n_tris(); m.n_verts = 1; m.n_tris = 1;
    m.has_n = m.has_s = m.has_uv = false;
    m.material = material_id; m.first_light = -1;
    m.reverse_orientation = flags & 1; m.swaps_handedness = o2w.swaps_handedness();
    m.alpha = 1.0f; m.shadow_alpha = 1.0f;
    m.sphere = (int)sc.spheres.size();
    sc.spheres.push_back(Sphere(o2w, (flags & 1) != 0, radius, z_min, z_max, phi_max));
    sc.P.push_back(o2w.point(V3(0, 0, 0))); sc.N.push_back(V3()); sc.S.push_back(V3()); sc.UV.push_back(V2());  // one vertex so that the slot's index triple is valid; never read
    const uint32_t mesh_id = (uint32_t)sc.meshes.size();
    for (int k = 0; k < 3; k++) sc.idx.push_back(m.vert_base);
    sc.tri_mesh.push_back(mesh_id);
    sc.meshes.push_back(m);
    sc.top_items.push_back(m.tri_base);
    s->built = false;
    return 0;
}
// AreaLightSource "diffuse" on the sphere added LAST (ORACLE ONLY): the sphere becomes the shape of a DiffuseAreaLight (lights/src/diffuse.rs) with radiance L
int oracle_make_last_sphere_a_light(OracleScene* s, const float L[3], int two_sided) {
    if (!s || !L) return -1;
    Scene& sc = s->sc;
    if (sc.spheres.empty() || sc.meshes.empty() || sc.meshes.back().sphere != (int)sc.spheres.size() - 1) { s->err = "the last shape is not a sphere"; return -2; }
    Light l{}; l.type = L_AREA; l.L = spec3(L); l.two_sided = two_sided; l.sphere = (int)sc.spheres.size() - 1; l.prim = sc.meshes.back().tri_base;
    const Sphere& sp = sc.spheres.back();
    l.area = sp.phi_max * sp.radius * (sp.z_max - sp.z_min);
    sc.meshes.back().first_light = (int32_t)sc.lights.size();
    sc.lights.push_back(l);
    s->built = false;
    return 0;
}
// Shape "hyperboloid" (hyperboloid.rs:392-416 reads p1, p2, phimax) — ORACLE ONLY
int oracle_add_hyperboloid(OracleScene* s, const float o2w_m[16], const float o2w_minv[16], const float p1[3], const float p2[3], float phi_max, uint32_t material_id, uint32_t flags) {
    if (!s || !o2w_m || !o2w_minv || !p1 || !p2) return -1;
    Scene& sc = s->sc;
    if (material_id >= sc.materials.size()) { s->err = "bad material id"; return -1; }
    if (sc.open_object >= 0) { s->err = "quadrics inside object instances are not restated"; return -5; }
    const Transform o2w(m4_from(o2w_m), m4_from(o2w_minv));
    Mesh m;
    m.vert_base = (uint32_t)sc.P.size(); m.tri_base = (uint32_t)sc.n_tris(); m.n_verts = 1; m.n_tris = 1;
    m.has_n = m.has_s = m.has_uv = false;
    m.material = material_id; m.first_light = -1;
    m.reverse_orientation = flags & 1; m.swaps_handedness = o2w.swaps_handedness();
    m.alpha = 1.0f; m.shadow_alpha = 1.0f;
    m.hyper = (int)sc.hyperboloids.size();
    sc.hyperboloids.push_back(Hyperboloid(o2w, (flags & 1) != 0, V3(p1[0], p1[1], p1[2]), V3(p2[0], p2[1], p2[2]), phi_max));
    sc.P.push_back(o2w.point(V3(0, 0, 0))); sc.N.push_back(V3()); sc.S.push_back(V3()); sc.UV.push_back(V2());
    const uint32_t mesh_id = (uint32_t)sc.meshes.size();
    for (int k = 0; k < 3; k++) sc.idx.push_back(m.vert_base);
    sc.tri_mesh.push_back(mesh_id);
    sc.meshes.push_back(m);
    sc.top_items.push_back(m.tri_base);
    s->built = false;
    return 0;
}
// Shape "cylinder" (radius, zmin, zmax) / "cone" (radius, height) / "paraboloid" (radius, zmin, zmax) / "disk" (radius, height, innerradius) — ORACLE ONLY.
// kind: 0 cylinder, 1 cone, 2 paraboloid, 3 disk; a, b: (zmin, zmax) / (height, -) / (zmin, zmax) / (height, innerradius)
int oracle_add_quadric(OracleScene* s, int kind, const float o2w_m[16], const float o2w_minv[16], float radius, float a, float b, float phi_max, uint32_t material_id, uint32_t flags) {
    if (!s || !o2w_m || !o2w_minv || kind < 0 || kind > 3) return -1;
    Scene& sc = s->sc;
    if (material_id >= sc.materials.size()) { s->err = "bad material id"; return -1; }
    if (sc.open_object >= 0) { s->err = "quadrics inside object instances are not restated"; return -5; }
    const Transform o2w(m4_from(o2w_m), m4_from(o2w_minv));
    Mesh m;
    m.vert_base = (uint32_t)sc.P.size(); m.tri_base = (uint32_t)sc.n_tris(); m.n_verts = 1; m.n_tris = 1;
    m.has_n = m.has_s = m.has_uv = false;
    m.material = material_id; m.first_light = -1;
    m.reverse_orientation = flags & 1; m.swaps_handedness = o2w.swaps_handedness();
    m.alpha = 1.0f; m.shadow_alpha = 1.0f;
    m.quadric = (int)sc.quadrics.size();
    sc.quadrics.push_back(Quadric(kind, o2w, (flags & 1) != 0, radius, a, b, phi_max));
    sc.P.push_back(o2w.point(V3(0, 0, 0))); sc.N.push_back(V3()); sc.S.push_back(V3()); sc.UV.push_back(V2());
    const uint32_t mesh_id = (uint32_t)sc.meshes.size();
    for (int k = 0; k < 3; k++) sc.idx.push_back(m.vert_base);
    sc.tri_mesh.push_back(mesh_id);
    sc.meshes.push_back(m);
    sc.top_items.push_back(m.tri_base);
    s->built = false;
    return 0;
}
// one ray against one sphere, outside any scene: out = {hit, t, p.xyz (world), n.xyz (world), u, v}
int oracle_sphere_probe(const float o2w_m[16], const float o2w_minv[16], float radius, float z_min, float z_max, float phi_max, uint32_t flags,
                        const float o[3], const float d[3], float t_max, float* out) {
    Scene sc; Renderer rr; rr.sc = &sc;
    const Transform o2w(m4_from(o2w_m), m4_from(o2w_minv));
    sc.spheres.push_back(Sphere(o2w, (flags & 1) != 0, radius, z_min, z_max, phi_max));
    Mesh m{}; m.sphere = 0; m.n_tris = 1; m.n_verts = 1; m.first_light = -1; m.alpha = m.shadow_alpha = 1.0f; m.alpha_tex = m.shadow_alpha_tex = -1;
    sc.meshes.push_back(m); sc.tri_mesh.push_back(0); sc.P.push_back(V3()); for (int k = 0; k < 3; k++) sc.idx.push_back(0);
    const Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), t_max, 0.0f);
    TriHit h;
    const bool hit = sc.sphere_intersect(r, sc.spheres[0], h);
    out[0] = hit ? 1.0f : 0.0f;
    if (!hit) return 0;
    const SurfaceHit si = rr.make_sphere_hit(r, 0, h, sc.spheres[0]);
    out[1] = h.t; out[2] = si.p.x; out[3] = si.p.y; out[4] = si.p.z; out[5] = si.n.x; out[6] = si.n.y; out[7] = si.n.z; out[8] = si.uv.x; out[9] = si.uv.y;
    out[10] = si.p_error.x; out[11] = si.p_error.y; out[12] = si.p_error.z;
    return 0;
}

// ObjectBegin / ObjectEnd / ObjectInstance (api/src/lib.rs:911-1000)
int oracle_object_begin(OracleScene* s, uint32_t* out_id) {
    if (!s || !out_id) return -1;
    Scene& sc = s->sc;
    if (sc.open_object >= 0) { s->err = "ObjectBegin called inside of an instance definition"; return -2; }
    Object ob; ob.tri0 = ob.tri1 = (uint32_t)sc.n_tris();
    sc.objects.push_back(ob);
    sc.open_object = (int)sc.objects.size() - 1;
    *out_id = (uint32_t)sc.open_object;
    return 0;
}
int oracle_object_end(OracleScene* s) {
    if (!s) return -1;
    if (s->sc.open_object < 0) { s->err = "ObjectEnd called outside of instance definition"; return -2; }
    s->sc.open_object = -1;
    return 0;
}
int oracle_add_instance(OracleScene* s, uint32_t object_id, const float i2w[16], const float w2i[16]) {
    if (!s || !i2w || !w2i) return -1;
    Scene& sc = s->sc;
    if (sc.open_object >= 0) { s->err = "ObjectInstance can't be called inside of instance definition"; return -2; }
    if (object_id >= sc.objects.size()) { s->err = "unknown object"; return -1; }
    if (sc.objects[object_id].tri1 == sc.objects[object_id].tri0) return 0;  // empty object: nothing is added (lib.rs:949-951)
    Instance in; in.object = object_id;
    M4 a, b; std::memcpy(a.m, i2w, 64); std::memcpy(b.m, w2i, 64);
    in.i2w = Transform(a, b);
    sc.instances.push_back(in);
    sc.top_items.push_back(ORC_INST_BIT | (uint32_t)(sc.instances.size() - 1));
    s->built = false;
    return 0;
}

int oracle_add_light_infinite(OracleScene* s, const float L[3], const float l2w[16], const float w2l[16]) {
    if (!s || !L || !l2w || !w2l) return -1;
    Light l{}; l.type = L_INFINITE; l.L = Spec(L[0], L[1], L[2]); l.l2w = Transform(m4_from(l2w), m4_from(w2l));
    infinite_light_setup(l);
    s->sc.infinite_lights.push_back((int)s->sc.lights.size());
    s->sc.lights.push_back(l);
    return 0;
}
// InfiniteAreaLight::new with a texmap (lights/src/infinite.rs:52-100): texels = image * L (no y flip here), MIPMap (EWA, repeat, 8), scalar image, Distribution2D
namespace {
int image_light_mipmap(OracleScene* s, int width, int height, const float* rgb) {   // MIPMap::new(.., Ewa, Repeat, 8.0) (projection.rs:70-80, goniometric.rs:49-58)
    if (!rgb) return -1;
    std::vector<Spec> tex((size_t)width * height);
    for (size_t i = 0; i < tex.size(); i++) tex[i] = Spec(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    MipMap m; m.filtering = TEX_FILTER_EWA; m.wrap = TEX_WRAP_REPEAT; m.max_anisotropy = 8.0f; m.is_float = false;
    m.build_from(tex, (size_t)width, (size_t)height);
    for (int i = 0; i < WEIGHT_LUT_SIZE; i++) { Float r2 = (Float)i / (Float)(WEIGHT_LUT_SIZE - 1); m.weight_lut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f); }
    s->sc.mipmaps.push_back(std::move(m));
    return (int)s->sc.mipmaps.size() - 1;
}
}  // namespace
// ProjectionLight::new (projection.rs:55-127); rgb may be null (no image: the light projects white inside its frustum, aspect 1)
int oracle_add_light_projection(OracleScene* s, const float I[3], const float l2w[16], const float w2l[16], float fov, int width, int height, const float* rgb) {
    if (!s || !I || !l2w || !w2l || (rgb && (width <= 0 || height <= 0))) return -1;
    Light l{}; l.type = L_PROJECTION; l.L = spec3(I); l.l2w = Transform(m4_from(l2w), m4_from(w2l)); l.prim = 0xFFFFFFFFu;
    l.p_light = l.l2w.point(V3(0, 0, 0));
    l.map_mip = image_light_mipmap(s, width, height, rgb);
    const Float aspect = rgb ? (Float)width / (Float)height : 1.0f;
    if (aspect > 1.0f) { l.screen[0] = -aspect; l.screen[1] = aspect; l.screen[2] = -1.0f; l.screen[3] = 1.0f; }
    else { l.screen[0] = -1.0f; l.screen[1] = 1.0f; l.screen[2] = -1.0f / aspect; l.screen[3] = 1.0f / aspect; }
    l.light_projection = t_perspective(fov, 1e-3f, 1e30f);
    V3 wc = normalize(l.light_projection.inv().point(V3(l.screen[1], l.screen[3], 0.0f)));
    l.cos_total_width = wc.z;
    s->sc.lights.push_back(l);
    return 0;
}
// GonioPhotometricLight::new (goniometric.rs:34-80)
int oracle_add_light_goniometric(OracleScene* s, const float I[3], const float l2w[16], const float w2l[16], int width, int height, const float* rgb) {
    if (!s || !I || !l2w || !w2l || (rgb && (width <= 0 || height <= 0))) return -1;
    Light l{}; l.type = L_GONIO; l.L = spec3(I); l.l2w = Transform(m4_from(l2w), m4_from(w2l)); l.prim = 0xFFFFFFFFu;
    l.p_light = l.l2w.point(V3(0, 0, 0));
    l.map_mip = image_light_mipmap(s, width, height, rgb);
    s->sc.lights.push_back(l);
    return 0;
}
int oracle_add_light_infinite_map(OracleScene* s, const float L[3], int width, int height, const float* rgb, const float l2w[16], const float w2l[16]) {
    if (!s || !L || !rgb || !l2w || !w2l || width <= 0 || height <= 0) return -1;
    Light l{}; l.type = L_INFINITE; l.L = spec3(L); l.l2w = Transform(m4_from(l2w), m4_from(w2l)); l.two_sided = 0; l.prim = 0xFFFFFFFFu; l.area = 0;
    infinite_light_setup(l);  // keeps the constant-light fields defined
    std::vector<Spec> tex((size_t)width * height);
    for (size_t i = 0; i < tex.size(); i++) tex[i] = Spec(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]) * l.L;
    MipMap m; m.filtering = TEX_FILTER_EWA; m.wrap = TEX_WRAP_REPEAT; m.max_anisotropy = 8.0f; m.is_float = false;
    m.build_from(tex, (size_t)width, (size_t)height);
    for (int i = 0; i < WEIGHT_LUT_SIZE; i++) { Float r2 = (Float)i / (Float)(WEIGHT_LUT_SIZE - 1); m.weight_lut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f); }
    const size_t dw = 2 * (size_t)m.pyr[0].w, dh = 2 * (size_t)m.pyr[0].h;   // compute_scalar_image (:326-369)
    const Float fwidth = 0.5f / (Float)(dw < dh ? dw : dh);
    l.dw = (int)dw; l.dh = (int)dh;
    l.d_cond_func.resize(dw * dh); l.d_cond_cdf.resize((dw + 1) * dh); l.d_cond_int.resize(dh);
    std::vector<Float> marg(dh);
    for (size_t v = 0; v < dh; v++) {
        Float vp = ((Float)v + 0.5f) / (Float)dh;
        Float sin_theta = std::sin(PI * ((Float)v + 0.5f) / (Float)dh);
        std::vector<Float> row(dw);
        for (size_t u = 0; u < dw; u++) { Float up = ((Float)u + 0.5f) / (Float)dw; row[u] = m.lookup_triangle_host(V2(up, vp), fwidth).y() * sin_theta; }
        Dist1D d; d.init(row);
        for (size_t u = 0; u < dw; u++) l.d_cond_func[v * dw + u] = d.func[u];
        for (size_t u = 0; u <= dw; u++) l.d_cond_cdf[v * (dw + 1) + u] = d.cdf[u];
        l.d_cond_int[v] = d.func_int; marg[v] = d.func_int;
    }
    Dist1D d; d.init(marg);
    l.d_marg_func = d.func; l.d_marg_cdf = d.cdf; l.d_marg_int = d.func_int;
    s->sc.mipmaps.push_back(std::move(m));
    l.map_mip = (int)s->sc.mipmaps.size() - 1;
    s->sc.infinite_lights.push_back((int)s->sc.lights.size());
    s->sc.lights.push_back(l);
    return 0;
}
int oracle_add_light_distant(OracleScene* s, const float L[3], const float w[3]) {
    if (!s || !L || !w) return -1;
    Light l{}; l.type = L_DISTANT; l.L = Spec(L[0], L[1], L[2]); l.w_light = V3(w[0], w[1], w[2]);
    s->sc.lights.push_back(l); return 0;
}
int oracle_add_light_point(OracleScene* s, const float I[3], const float p[3]) {
    if (!s || !I || !p) return -1;
    Light l{}; l.type = L_POINT; l.L = Spec(I[0], I[1], I[2]); l.p_light = V3(p[0], p[1], p[2]);
    s->sc.lights.push_back(l); return 0;
}
int oracle_add_light_spot(OracleScene* s, const float I[3], const float l2w[16], const float w2l[16], float cos_total_width, float cos_falloff_start) {  // spot.rs:27-50
    if (!s || !I || !l2w || !w2l) return -1;
    Light l{}; l.type = L_SPOT; l.L = Spec(I[0], I[1], I[2]);
    M4 a, b; std::memcpy(a.m, l2w, 64); std::memcpy(b.m, w2l, 64);
    l.l2w = Transform(a, b);
    l.p_light = l.l2w.point(V3(0, 0, 0));
    l.cos_total_width = cos_total_width; l.cos_falloff_start = cos_falloff_start;
    s->sc.lights.push_back(l);
    return 0;
}
int oracle_add_light_diffuse_area(OracleScene* s, const float L[3], int two_sided, uint32_t n_tris, uint32_t* out_first) {
    if (!s || !L) return -1;
    if (out_first) *out_first = (uint32_t)s->sc.lights.size();
    for (uint32_t i = 0; i < n_tris; i++) {
        Light l{}; l.type = L_AREA; l.L = Spec(L[0], L[1], L[2]); l.two_sided = two_sided; l.prim = 0xffffffffu; l.area = 0;
        s->sc.lights.push_back(l);
    }
    return 0;
}
int oracle_set_camera_perspective(OracleScene* s, const float r2c[16], const float c2w[16], float lens_radius, float focal_distance,
                                  float shutter_open, float shutter_close) {
    if (!s || !r2c || !c2w) return -1;
    s->r.cam.raster_to_camera = Transform(m4_from(r2c), M4::identity());  // only .m is used on the path
    s->r.cam.camera_to_world = Transform(m4_from(c2w), M4::identity());
    s->r.cam.lens_radius = lens_radius; s->r.cam.focal_distance = focal_distance;
    s->r.cam.shutter_open = shutter_open; s->r.cam.shutter_close = shutter_close;
    const Transform& rc = s->r.cam.raster_to_camera;  // perspective_camera.rs:70-74
    s->r.cam.dx_camera = rc.point(V3(1, 0, 0)) - rc.point(V3(0, 0, 0));
    s->r.cam.dy_camera = rc.point(V3(0, 1, 0)) - rc.point(V3(0, 0, 0));
    s->r.cam.kind = 0;
    s->have_camera = true; return 0;
}
int oracle_set_camera_orthographic(OracleScene* s, const float r2c[16], const float c2w[16], float lens_radius, float focal_distance,
                                   float shutter_open, float shutter_close) {  // OrthographicCamera::new (orthographic_camera.rs:39-72)
    const int rc_ = oracle_set_camera_perspective(s, r2c, c2w, lens_radius, focal_distance, shutter_open, shutter_close);
    if (rc_) return rc_;
    const Transform& rc = s->r.cam.raster_to_camera;
    s->r.cam.dx_camera = rc.vector(V3(1, 0, 0)); s->r.cam.dy_camera = rc.vector(V3(0, 1, 0));   // :58-64: transform_vector, not a difference of points
    s->r.cam.kind = 1;
    return 0;
}

int oracle_set_camera_environment(OracleScene* s, const float c2w[16], int xres, int yres, float shutter_open, float shutter_close) {  // environment_camera.rs:27-41
    if (!s || !c2w || xres <= 0 || yres <= 0) return -1;
    s->r.cam = Camera();
    s->r.cam.camera_to_world = Transform(m4_from(c2w), M4::identity());
    s->r.cam.shutter_open = shutter_open; s->r.cam.shutter_close = shutter_close;
    s->r.cam.full_res[0] = (Float)xres; s->r.cam.full_res[1] = (Float)yres;
    s->r.cam.kind = 2;
    s->have_camera = true; return 0;
}

// ---- textures (oracle_texture.hpp).  One id space for float and spectrum textures; float ones carry three equal channels.
int oracle_add_mipmap(OracleScene* s, int width, int height, const float* rgb, int as_float, float scale, int gamma, int filtering, int wrap,
                      float max_anisotropy, uint32_t* out_id) {
    if (!s || !rgb || width <= 0 || height <= 0) return -1;
    MipMap m; m.build(rgb, (size_t)width, (size_t)height, as_float != 0, scale, gamma != 0, filtering, wrap, max_anisotropy);
    s->sc.mipmaps.push_back(std::move(m));
    if (out_id) *out_id = (uint32_t)s->sc.mipmaps.size() - 1;
    return 0;
}
static int push_texture(OracleScene* s, const Texture& t, uint32_t* out_id) {
    s->sc.textures.push_back(t);
    if (out_id) *out_id = (uint32_t)s->sc.textures.size() - 1;
    return 0;
}
int oracle_add_texture_constant(OracleScene* s, const float v[3], uint32_t* out_id) {
    if (!s || !v) return -1;
    Texture t; t.kind = TK_CONST; t.c = spec3(v); return push_texture(s, t, out_id);
}
int oracle_add_texture_scale(OracleScene* s, uint32_t t1, uint32_t t2, uint32_t* out_id) {
    if (!s || t1 >= s->sc.textures.size() || t2 >= s->sc.textures.size()) return -1;
    Texture t; t.kind = TK_SCALE; t.t1 = (int)t1; t.t2 = (int)t2; return push_texture(s, t, out_id);
}
int oracle_add_texture_mix(OracleScene* s, uint32_t t1, uint32_t t2, uint32_t amount, uint32_t* out_id) {
    if (!s || t1 >= s->sc.textures.size() || t2 >= s->sc.textures.size() || amount >= s->sc.textures.size()) return -1;
    Texture t; t.kind = TK_MIX; t.t1 = (int)t1; t.t2 = (int)t2; t.amount = (int)amount; return push_texture(s, t, out_id);
}
int oracle_add_texture_imagemap(OracleScene* s, uint32_t mipmap, float su, float sv, float du, float dv, uint32_t* out_id) {
    if (!s || mipmap >= s->sc.mipmaps.size()) return -1;
    Texture t; t.kind = TK_IMAGE; t.mip = (int)mipmap; t.su = su; t.sv = sv; t.du = du; t.dv = dv; return push_texture(s, t, out_id);
}
int oracle_add_texture_checkerboard(OracleScene* s, uint32_t t1, uint32_t t2, float su, float sv, float du, float dv, int aa_mode, uint32_t* out_id) {
    if (!s || t1 >= s->sc.textures.size() || t2 >= s->sc.textures.size()) return -1;
    Texture t; t.kind = TK_CHECKER; t.t1 = (int)t1; t.t2 = (int)t2; t.su = su; t.sv = sv; t.du = du; t.dv = dv; t.aa = aa_mode; return push_texture(s, t, out_id);
}
int oracle_add_texture_uv(OracleScene* s, float su, float sv, float du, float dv, uint32_t* out_id) {
    if (!s) return -1;
    Texture t; t.kind = TK_UV; t.su = su; t.sv = sv; t.du = du; t.dv = dv; return push_texture(s, t, out_id);
}
int oracle_add_texture_bilerp(OracleScene* s, const float v00[3], const float v01[3], const float v10[3], const float v11[3], float su, float sv, float du, float dv,
                              uint32_t* out_id) {
    if (!s || !v00 || !v01 || !v10 || !v11) return -1;
    Texture t; t.kind = TK_BILERP; t.v[0] = spec3(v00); t.v[1] = spec3(v01); t.v[2] = spec3(v10); t.v[3] = spec3(v11); t.su = su; t.sv = sv; t.du = du; t.dv = dv;
    return push_texture(s, t, out_id);
}
int oracle_add_texture_dots(OracleScene* s, uint32_t inside, uint32_t outside, float su, float sv, float du, float dv, uint32_t* out_id) {
    if (!s || inside >= s->sc.textures.size() || outside >= s->sc.textures.size()) return -1;
    Texture t; t.kind = TK_DOTS; t.t1 = (int)inside; t.t2 = (int)outside; t.su = su; t.sv = sv; t.du = du; t.dv = dv; return push_texture(s, t, out_id);
}
int oracle_set_texture_mapping(OracleScene* s, uint32_t tex, int kind, const float* prm) {
    if (!s || tex >= s->sc.textures.size() || kind < 1 || kind > 3 || !prm) return -1;
    Texture& t = s->sc.textures[tex];
    if (!(t.kind == TK_IMAGE || t.kind == TK_CHECKER || t.kind == TK_UV || t.kind == TK_BILERP || t.kind == TK_DOTS)) return -6;
    t.mapping = kind;
    if (kind == 3) { t.vs = V3(prm[0], prm[1], prm[2]); t.vt = V3(prm[3], prm[4], prm[5]); t.du = prm[6]; t.dv = prm[7]; }
    else t.w2t = m4_from(prm);
    return 0;
}
// probes for the pinning tests
static int push_texture3d(OracleScene* s, int kind, const float m[16], float omega, int octaves, float scale, float variation, uint32_t t1, uint32_t t2, uint32_t* out_id) {
    if (!s || !m) return -1;
    Texture t; t.kind = kind; t.w2t = m4_from(m); t.omega = omega; t.octaves = octaves; t.scale = scale; t.variation = variation; t.t1 = (int)t1; t.t2 = (int)t2;
    return push_texture(s, t, out_id);
}
int oracle_add_texture_fbm(OracleScene* s, const float m[16], float omega, int octaves, uint32_t* out_id) { return push_texture3d(s, TK_FBM, m, omega, octaves, 1, 0, 0, 0, out_id); }
int oracle_add_texture_wrinkled(OracleScene* s, const float m[16], float omega, int octaves, uint32_t* out_id) { return push_texture3d(s, TK_WRINKLED, m, omega, octaves, 1, 0, 0, 0, out_id); }
int oracle_add_texture_windy(OracleScene* s, const float m[16], uint32_t* out_id) { return push_texture3d(s, TK_WINDY, m, 0.5f, 0, 1, 0, 0, 0, out_id); }
int oracle_add_texture_marble(OracleScene* s, const float m[16], float omega, int octaves, float scale, float variation, uint32_t* out_id) {
    return push_texture3d(s, TK_MARBLE, m, omega, octaves, scale, variation, 0, 0, out_id);
}
int oracle_add_texture_checkerboard3d(OracleScene* s, uint32_t t1, uint32_t t2, const float m[16], uint32_t* out_id) {
    if (!s || t1 >= s->sc.textures.size() || t2 >= s->sc.textures.size()) return -1;
    return push_texture3d(s, TK_CHECKER3D, m, 0, 0, 1, 0, t1, t2, out_id);
}
int oracle_texture_eval_batch(OracleScene* s, uint32_t tex, uint64_t n, const float* in /*15 per point: u v dudx dvdx dudy dvdy p[3] dpdx[3] dpdy[3]*/, float* out_rgb) {
    if (!s || tex >= s->sc.textures.size()) return -1;
    for (uint64_t i = 0; i < n; i++) {
        const float* q = in + 15 * i;
        TexCtx c; c.uv = V2(q[0], q[1]); c.dudx = q[2]; c.dvdx = q[3]; c.dudy = q[4]; c.dvdy = q[5];
        c.p = V3(q[6], q[7], q[8]); c.dpdx = V3(q[9], q[10], q[11]); c.dpdy = V3(q[12], q[13], q[14]);
        Spec v = tex_eval(s->sc.textures, s->sc.mipmaps, (int)tex, c);
        out_rgb[3 * i] = v.c[0]; out_rgb[3 * i + 1] = v.c[1]; out_rgb[3 * i + 2] = v.c[2];
    }
    return 0;
}
int oracle_mipmap_levels(OracleScene* s, uint32_t mip, int* out_levels, int* out_wh /*2 per level, up to 32 levels*/) {
    if (!s || mip >= s->sc.mipmaps.size()) return -1;
    const MipMap& m = s->sc.mipmaps[mip];
    *out_levels = (int)m.pyr.size();
    for (size_t i = 0; i < m.pyr.size() && i < 32; i++) { out_wh[2 * i] = (int)m.pyr[i].w; out_wh[2 * i + 1] = (int)m.pyr[i].h; }
    return 0;
}
int oracle_mipmap_level_texels(OracleScene* s, uint32_t mip, int level, float* out_rgb) {
    if (!s || mip >= s->sc.mipmaps.size() || level < 0 || (size_t)level >= s->sc.mipmaps[mip].pyr.size()) return -1;
    const MipMap::Level& l = s->sc.mipmaps[mip].pyr[(size_t)level];
    for (size_t i = 0; i < l.t.size(); i++) { out_rgb[3 * i] = l.t[i].c[0]; out_rgb[3 * i + 1] = l.t[i].c[1]; out_rgb[3 * i + 2] = l.t[i].c[2]; }
    return 0;
}
// UberMaterial with an opacity texture: the lobe list becomes every lobe the material CAN have (uber.rs:126-160), colours decided per hit
static void uber_rebuild_for_opacity(Material& m) {
    if (m.rebuilt) return;
    const std::vector<Lobe> old = m.lobes;
    int oldp[4]; for (int k = 0; k < 4; k++) oldp[k] = m.param_lobe[k];
    const int old_rough = m.rough_lobe;
    m.lobes.clear();
    for (int k = 0; k < 4; k++) { m.param_lobe[k] = -1; m.param_field[k] = 0; }
    m.rough_lobe = -1;
    { Lobe l; l.kind = LK_SPEC_T; l.type = BX_TRANS | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.0f; l.pre_mode = 4; m.lobes.push_back(l); }
    const Float e = m.raw_eta;
    for (int k = 0; k < 4; k++) {
        const Spec base = spec_clamp0(m.raw_k[k]);
        const Lobe* ol = oldp[k] >= 0 ? &old[(size_t)oldp[k]] : nullptr;
        const int tex = ol ? (k == 3 ? ol->t_tex : ol->r_tex) : -1;
        if (base.is_black() && tex < 0) continue;   // op * 0 is black at every hit: the lobe is never added
        Lobe l; l.pre_mode = 3; l.pre = base;
        if (k == 0) { l.kind = LK_LAMBERT; l.type = BX_REFL | BX_DIFF; l.r_tex = tex; }
        else if (k == 1) {
            l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r_tex = tex;
            Float ur = m.raw_ur, vr = m.raw_vr;
            if (m.raw_remap) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
            set_tr(l, ur, vr);
            if (old_rough >= 0) { const Lobe& r = old[(size_t)old_rough]; l.ax_tex = r.ax_tex; l.ay_tex = r.ay_tex; l.remap = r.remap; }
            m.rough_lobe = (int)m.lobes.size();
        }
        else if (k == 2) { l.kind = LK_SPEC_R; l.type = BX_REFL | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r_tex = tex; }
        else { l.kind = LK_SPEC_T; l.type = BX_TRANS | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.t_tex = tex; m.param_field[3] = 1; }
        m.param_lobe[k] = (int)m.lobes.size();
        m.lobes.push_back(l);
    }
    m.has_pre = false;   // the per-hit opacity takes the constant's place
    m.bsdf_eta_alt = e;
    m.rebuilt = true;
}
// GlassMaterial with a roughness texture: which lobes a hit gets depends on `urough == 0 && vrough == 0` there (glass.rs:110-141)
static void glass_rebuild_for_roughness(Material& m) {
    if (m.rebuilt) return;
    const std::vector<Lobe> old = m.lobes;
    int rtex = -1, ttex = -1;
    for (const Lobe& l : old) { if (l.r_tex >= 0) rtex = l.r_tex; if (l.t_tex >= 0) ttex = l.t_tex; }
    const Spec r = spec_clamp0(m.raw_k[2]), t = spec_clamp0(m.raw_k[3]);
    m.lobes.clear();
    for (int k = 0; k < 4; k++) { m.param_lobe[k] = m.param_lobe2[k] = -1; m.param_field[k] = m.param_field2[k] = 0; }
    m.rough_lobe = m.rough_lobe2 = -1;
    const bool has_r = !r.is_black() || rtex >= 0, has_t = !t.is_black() || ttex >= 0;
    if (has_r || has_t) {
        Float ur = m.raw_ur, vr = m.raw_vr;
        if (m.raw_remap) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        Lobe f; f.kind = LK_FRESNEL_SPEC; f.type = BX_REFL | BX_TRANS | BX_SPEC; f.r = r; f.t = t; f.r_tex = rtex; f.t_tex = ttex; f.eta_a = 1.0f; f.eta_b = m.raw_eta;
        f.alt = 1; f.ur_raw = m.raw_ur; f.vr_raw = m.raw_vr; set_tr(f, ur, vr);
        m.param_lobe[2] = 0; m.param_field[2] = 0; m.param_lobe[3] = 0; m.param_field[3] = 1;
        m.lobes.push_back(f);
        if (has_r) { Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = m.raw_eta; l.r = r; l.r_tex = rtex; l.alt = 2; l.ur_raw = m.raw_ur; l.vr_raw = m.raw_vr;
                     set_tr(l, ur, vr); m.param_lobe2[2] = (int)m.lobes.size(); m.param_field2[2] = 0; m.lobes.push_back(l); }
        if (has_t) { Lobe l; l.kind = LK_MICRO_T; l.type = BX_TRANS | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = m.raw_eta; l.t = t; l.t_tex = ttex; l.alt = 2; l.ur_raw = m.raw_ur; l.vr_raw = m.raw_vr;
                     set_tr(l, ur, vr); m.param_lobe2[3] = (int)m.lobes.size(); m.param_field2[3] = 1; m.lobes.push_back(l); }
    }
    m.rebuilt = true;
}
// TranslucentMaterial with a reflect / transmit texture: every lobe the material can have (translucent.rs:76-98), colours and presence decided per hit
static void translucent_rebuild_rt(Material& m) {
    if (m.rt_mode) return;
    const std::vector<Lobe> old = m.lobes;
    int btex[2] = {-1, -1};   // Kd / Ks textures already set
    for (int k = 0; k < 2; k++) if (m.param_lobe[k] >= 0) { const Lobe& l = old[(size_t)m.param_lobe[k]]; btex[k] = m.param_field[k] == 0 ? l.r_tex : l.t_tex; }
    int axt = -1, ayt = -1;
    if (m.rough_lobe >= 0) { axt = old[(size_t)m.rough_lobe].ax_tex; ayt = old[(size_t)m.rough_lobe].ay_tex; }
    m.lobes.clear();
    for (int k = 0; k < 4; k++) { m.param_lobe[k] = m.param_lobe2[k] = -1; m.param_field[k] = m.param_field2[k] = 0; }
    m.rough_lobe = m.rough_lobe2 = -1;
    for (int k = 0; k < 2; k++) {
        if (m.raw_k[k].is_black() && btex[k] < 0) continue;   // Kd / Ks black at every hit: `if !kd.is_black()` never holds
        for (int side = 0; side < 2; side++) {
            Lobe l;
            if (k == 0) { l.kind = side ? LK_LAMBERT_T : LK_LAMBERT; l.type = (side ? BX_TRANS : BX_REFL) | BX_DIFF; }
            else {
                l.kind = side ? LK_MICRO_T : LK_MICRO_R; l.type = (side ? BX_TRANS : BX_REFL) | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f;
                const Float rough = m.raw_remap ? roughness_to_alpha(m.raw_ur) : m.raw_ur;
                set_tr(l, rough, rough);
                l.ax_tex = axt; l.ay_tex = ayt; l.remap = m.raw_remap;
                (side ? m.rough_lobe2 : m.rough_lobe) = (int)m.lobes.size();
            }
            l.pre_mode = 5; l.pre = m.raw_k[k];
            (side ? l.t_tex : l.r_tex) = btex[k];
            (side ? m.param_lobe2[k] : m.param_lobe[k]) = (int)m.lobes.size(); (side ? m.param_field2[k] : m.param_field[k]) = side;
            m.lobes.push_back(l);
        }
    }
    m.rough_remap = m.raw_remap;
    m.none = false; m.rt_mode = true;
}
// param: 0 Kd, 1 Ks, 2 Kr, 3 Kt; 4 uber opacity, 5 mix amount, 6 metal eta, 7 metal k, 8 translucent reflect, 9 translucent transmit
int oracle_set_material_texture(OracleScene* s, uint32_t material, int param, uint32_t texture) {
    if (!s || material >= s->sc.materials.size() || texture >= s->sc.textures.size() || param < 0 || param > 9) return -1;
    Material& m = s->sc.materials[material];
    if (param >= 8) {   // translucent.rs:70-71
        if (m.made_as != 5) return -6;
        translucent_rebuild_rt(m);
        (param == 8 ? m.refl_tex : m.trans_tex) = (int)texture; m.textured = true;
        return 0;
    }
    if (param == 4) {
        if (m.made_as != 1) return -6;
        uber_rebuild_for_opacity(m);
        m.opacity_tex = (int)texture; m.textured = true;
        return 0;
    }
    if (param == 5) {
        if (m.made_as != 4) return -6;
        {   // the product carries at most six per-hit colours per material (its texture pass hands them to the shade pass in six slots): refuse what it refuses
            int cols = 1;
            for (const Lobe& l : m.lobes)
                cols += ((l.r_tex >= 0 || (l.pre_mode == 3 && l.kind != LK_SPEC_T)) ? 1 : 0) + ((l.t_tex >= 0 || (l.pre_mode == 3 && l.kind == LK_SPEC_T) || l.pre_mode == 4) ? 1 : 0) +
                        (l.eta_tex >= 0 ? 1 : 0) + (l.k_tex >= 0 ? 1 : 0);
            if (cols > 6) { s->err = "set_material_texture: a mix with a textured amount may have at most 5 more per-hit colours"; return -5; }
        }
        for (size_t i = 0; i < m.lobes.size(); i++) { Lobe& l = m.lobes[i]; l.amt_side = (int)i < m.mix_n1 ? 1 : 2; l.amt_level = l.n_scale - 1; }
        m.amount_tex = (int)texture; m.textured = true;
        return 0;
    }
    if (param == 6 || param == 7) {
        if (m.made_as != 3) return -6;
        (param == 6 ? m.lobes[0].eta_tex : m.lobes[0].k_tex) = (int)texture; m.textured = true;
        return 0;
    }
    if (m.param_lobe[param] < 0) return -6;
    const int lobes[2] = {m.param_lobe[param], m.param_lobe2[param]}, fields[2] = {m.param_field[param], m.param_field2[param]};
    for (int k = 0; k < 2; k++) if (lobes[k] >= 0) {
        Lobe& l = m.lobes[(size_t)lobes[k]];
        if (fields[k] == 0) l.r_tex = (int)texture; else l.t_tex = (int)texture;
        if (m.has_pre) { l.has_pre = true; l.pre = m.pre; }
        if (l.pre_raw_test) l.has_pre = true;
    }
    m.textured = true;
    return 0;
}
int oracle_set_last_mesh_alpha_textures(OracleScene* s, uint32_t alpha_tex, uint32_t shadow_alpha_tex) {
    if (!s || s->sc.meshes.empty()) return -1;
    if ((alpha_tex != 0xFFFFFFFFu && alpha_tex >= s->sc.textures.size()) || (shadow_alpha_tex != 0xFFFFFFFFu && shadow_alpha_tex >= s->sc.textures.size())) return -1;
    Mesh& m = s->sc.meshes.back();
    if (alpha_tex != 0xFFFFFFFFu) m.alpha_tex = (int)alpha_tex;
    if (shadow_alpha_tex != 0xFFFFFFFFu) m.shadow_alpha_tex = (int)shadow_alpha_tex;
    return 0;
}
int oracle_set_material_float_texture(OracleScene* s, uint32_t material, int fparam, uint32_t texture) {  // 0 sigma, 1 uroughness, 2 vroughness, 3 index
    if (!s || material >= s->sc.materials.size() || texture >= s->sc.textures.size() || fparam < 0 || fparam > 3) return -1;
    if (fparam == 3) {   // glass.rs:102, uber.rs:128
        if (s->sc.materials[material].made_as != 1 && s->sc.materials[material].made_as != 2) return -6;
        if (s->sc.materials[material].made_as == 1 && s->sc.materials[material].opacity_tex < 0) {   // uber: everything about the hit's lobe list becomes per hit (as in the reference)
            const Material mq = s->sc.materials[material];
            const float op[3] = {mq.has_pre ? mq.pre.c[0] : 1.0f, mq.has_pre ? mq.pre.c[1] : 1.0f, mq.has_pre ? mq.pre.c[2] : 1.0f};
            uint32_t op_tex = 0;
            if (oracle_add_texture_constant(s, op, &op_tex) != 0) return -1;
            const int rc = oracle_set_material_texture(s, material, 4, op_tex);
            if (rc) return rc;
        }
        Material& mi = s->sc.materials[material];
        if (mi.lobes.empty()) return -6;
        mi.index_tex = (int)texture; mi.textured = true;
        return 0;
    }
    Material& m = s->sc.materials[material];
    if (fparam == 0) {
        if (m.general || m.none || m.lobes.size() != 1 || !(m.lobes[0].kind == LK_LAMBERT || m.lobes[0].kind == LK_OREN)) return -6;
        m.lobes[0].sigma_tex = (int)texture;
    } else if (m.made_as == 2) {  // glass: every alternative lobe carries the textures (the smooth one for its `== 0` test)
        glass_rebuild_for_roughness(m);
        if (m.lobes.empty()) return -6;
        for (Lobe& l : m.lobes) { (fparam == 1 ? l.ax_tex : l.ay_tex) = (int)texture; l.remap = m.raw_remap; }
    } else {
        if (m.rough_lobe < 0) return -6;
        const int rl[2] = {m.rough_lobe, m.rough_lobe2};
        for (int k = 0; k < 2; k++) if (rl[k] >= 0) { Lobe& l = m.lobes[(size_t)rl[k]]; (fparam == 1 ? l.ax_tex : l.ay_tex) = (int)texture; l.remap = m.rough_remap; }
    }
    m.textured = true;
    return 0;
}
int oracle_set_material_bump(OracleScene* s, uint32_t material, uint32_t texture) {
    if (!s || material >= s->sc.materials.size() || texture >= s->sc.textures.size()) return -1;
    if (s->sc.materials[material].none) return -6;
    s->sc.materials[material].bump_tex = (int)texture; return 0;
}
int oracle_add_material_matte_tex(OracleScene* s, uint32_t kd_tex, float sigma, uint32_t* out_id) {  // matte.rs:47-76 with a texture for Kd
    const float one[3] = {1.0f, 1.0f, 1.0f};
    uint32_t id = 0;
    int rc = oracle_add_material_matte(s, one, sigma, &id);
    if (rc == 0) rc = oracle_set_material_texture(s, id, 0, kd_tex);
    if (rc == 0 && out_id) *out_id = id;
    return rc;
}
int oracle_set_film(OracleScene* s, int xres, int yres, const int crop[4], const float radius[2], const float table[256], float scale,
                    float max_lum) {
    if (!s || !crop || !radius || !table) return -1;
    FilmCfg& f = s->r.film; f.xres = xres; f.yres = yres;
    for (int i = 0; i < 4; i++) f.crop[i] = crop[i];
    f.radius[0] = radius[0]; f.radius[1] = radius[1];
    std::memcpy(f.table, table, sizeof(f.table)); f.scale = scale; f.max_lum = max_lum;
    s->have_film = true; return 0;
}
int oracle_set_sampler(OracleScene* s, int kind, uint32_t spp, const int sb[4], int at_center) {
    if (!s || !sb) return -1;
    SamplerConfig& c = s->r.scfg; c.kind = kind; c.spp = spp; c.at_center = at_center != 0;
    for (int i = 0; i < 4; i++) c.bounds[i] = sb[i];
    s->have_sampler = true; return 0;
}
int oracle_set_sobol_tables(OracleScene* s, const uint32_t* m32, size_t n32, const uint64_t* vdc, const uint64_t* vdc_inv, size_t n_each) {
    if (!s || !m32 || !vdc || !vdc_inv) return -1;
    s->sobol32.assign(m32, m32 + n32); s->vdc.assign(vdc, vdc + n_each); s->vdc_inv.assign(vdc_inv, vdc_inv + n_each);
    s->r.scfg.sobol.m32 = s->sobol32.data(); s->r.scfg.sobol.vdc = s->vdc.data(); s->r.scfg.sobol.vdc_inv = s->vdc_inv.data();
    return 0;
}
int oracle_build_accel(OracleScene* s, int split_method, int max_prims) {
    if (!s) return -1;
    s->sc.build_bvh(split_method, max_prims);
    if (split_method == 1 && s->sc.hlbvh_panic) { s->err = "HLBVH build hit one of the reference's assertions (hlbvh.rs:338/356/418, mod.rs:128)"; return -1; }
    s->built = true; return 0;
}
int oracle_world_bound(const OracleScene* s, float out[6]) {
    if (!s || !s->built) return -2;
    const Bounds3& b = s->sc.world_bound;
    out[0] = b.pmin.x; out[1] = b.pmin.y; out[2] = b.pmin.z; out[3] = b.pmax.x; out[4] = b.pmax.y; out[5] = b.pmax.z;
    return 0;
}

static int batch_threads() { const char* e = std::getenv("ORACLE_THREADS"); int n = e ? std::atoi(e) : 0; if (n <= 0) n = (int)std::thread::hardware_concurrency(); return n > 0 ? n : 1; }

int oracle_intersect_batch_stats(OracleScene* s, const OracleRay* rays, OracleHit* hits, uint64_t n, OracleTraversalStats* st, int n_threads) {
    if (!s || (!rays && n) || (!hits && n)) return -1;
    if (!s->built) { s->err = "build_accel first"; return -2; }
    if (n_threads <= 0) n_threads = batch_threads();
    std::vector<TraversalStats> ts(n_threads);
    auto work = [&](int tid) {
        for (uint64_t i = tid; i < n; i += n_threads) {
            Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].t_max, rays[i].time);
            uint32_t prim = 0xffffffffu; TriHit h{0, 0, 0, 0};
            bool found = s->sc.intersect(r, prim, h, st ? &ts[tid] : nullptr);
            OracleHit& o = hits[i]; std::memset(&o, 0, sizeof(o));
            if (found) { o.t = h.t; o.prim = prim; o.b0 = h.b0; o.b1 = h.b1; o.b2 = h.b2; o.pad[1] = h.inst; }
            else { o.t = rays[i].t_max; o.prim = 0xffffffffu; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    if (st) { st->rays = st->nodes_visited = st->tri_tests = 0; for (auto& t : ts) { st->rays += t.rays; st->nodes_visited += t.nodes_visited; st->tri_tests += t.tri_tests; } }
    return 0;
}
int oracle_intersect_batch(OracleScene* s, const OracleRay* rays, OracleHit* hits, uint64_t n) {
    return oracle_intersect_batch_stats(s, rays, hits, n, nullptr, 0);
}
int oracle_occluded_batch_stats(OracleScene* s, const OracleRay* rays, uint8_t* out, uint64_t n, OracleTraversalStats* st, int n_threads) {
    if (!s || (!rays && n) || (!out && n)) return -1;
    if (!s->built) { s->err = "build_accel first"; return -2; }
    if (n_threads <= 0) n_threads = batch_threads();
    std::vector<TraversalStats> ts(n_threads);
    auto work = [&](int tid) {
        for (uint64_t i = tid; i < n; i += n_threads) {
            Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].t_max, rays[i].time);
            out[i] = s->sc.intersect_p(r, st ? &ts[tid] : nullptr) ? 1 : 0;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    if (st) { st->rays = st->nodes_visited = st->tri_tests = 0; for (auto& t : ts) { st->rays += t.rays; st->nodes_visited += t.nodes_visited; st->tri_tests += t.tri_tests; } }
    return 0;
}
int oracle_occluded_batch(OracleScene* s, const OracleRay* rays, uint8_t* out, uint64_t n) {
    return oracle_occluded_batch_stats(s, rays, out, n, nullptr, 0);
}

// n_threads <= 0: all host cores.  record_cap > 0: also capture up to that many regular and shadow rays.
int oracle_render_path_ex(OracleScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size,
                          int tile_part, int tile_parts, float* out_xyz, float* out_weight, OracleStats* st, int n_threads,
                          int count_traversal, uint64_t* out_nv_nt /*4: nv_reg, nt_reg, nv_sh, nt_sh*/) {
    if (!s || !pixel_bounds || !out_xyz || !out_weight) return -1;
    if (!s->built || !s->have_camera || !s->have_film || !s->have_sampler) { s->err = "scene incomplete"; return -2; }
    if (s->r.scfg.kind == 1 && !s->r.scfg.sobol.m32) { s->err = "sobol tables not set"; return -2; }
    Renderer& r = s->r;
    r.max_depth = max_depth; r.rr_threshold = rr_threshold; r.light_strategy = light_strategy;
    for (int i = 0; i < 4; i++) r.pixel_bounds[i] = pixel_bounds[i];
    r.count_traversal = count_traversal != 0;
    r.rec = s->rec.cap ? &s->rec : nullptr;
    if (n_threads <= 0) n_threads = batch_threads();
    auto t0 = std::chrono::steady_clock::now();
    r.render(tile_size, tile_part, tile_parts, n_threads, out_xyz, out_weight);
    auto t1 = std::chrono::steady_clock::now();
    if (st) {
        std::memset(st, 0, sizeof(*st));
        st->camera_rays = r.total_stats.camera_rays; st->regular_rays = r.total_stats.regular_rays; st->shadow_rays = r.total_stats.shadow_rays;
        st->paths_zero_radiance = r.total_stats.zero_paths; st->paths_total = r.total_stats.total_paths;
        st->render_seconds = std::chrono::duration<double>(t1 - t0).count();
        st->light_distributions_created = r.spatial ? r.spatial_created : 0;
    }
    if (out_nv_nt) { out_nv_nt[0] = r.total_stats.nv_regular; out_nv_nt[1] = r.total_stats.nt_regular; out_nv_nt[2] = r.total_stats.nv_shadow; out_nv_nt[3] = r.total_stats.nt_shadow; }
    return 0;
}
int oracle_add_material_substrate(OracleScene* s, const float kd[3], const float ks[3], float urough, float vrough, int remap, uint32_t* out_id) {  // substrate.rs:55-84
    if (!s || !kd || !ks) return -1;
    Material m; m.general = true;
    Spec d = spec_clamp0(spec3(kd)), sp = spec_clamp0(spec3(ks));
    if (!d.is_black() || !sp.is_black()) {
        if (remap) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
        Lobe l; l.kind = LK_FRESNEL_BLEND; l.type = BX_REFL | BX_GLOSSY; l.r = d; l.t = sp; set_tr(l, urough, vrough);
        m.rough_lobe = (int)m.lobes.size(); m.rough_remap = remap != 0;
        m.lobes.push_back(l);
        m.param_lobe[0] = 0; m.param_field[0] = 0; m.param_lobe[1] = 0; m.param_field[1] = 1;
    }
    return push_material(s, m, out_id);
}
int oracle_add_material_translucent(OracleScene* s, const float kd[3], const float ks[3], const float reflect[3], const float transmit[3], float roughness, int remap,
                                    uint32_t* out_id) {  // translucent.rs:57-112
    if (!s || !kd || !ks || !reflect || !transmit) return -1;
    Material m; m.general = true; m.bsdf_eta = 1.5f;
    Spec r = spec_clamp0(spec3(reflect)), t = spec_clamp0(spec3(transmit));
    m.made_as = 5; m.raw_k[0] = spec_clamp0(spec3(kd)); m.raw_k[1] = spec_clamp0(spec3(ks)); m.raw_k[2] = r; m.raw_k[3] = t; m.raw_ur = roughness; m.raw_remap = remap != 0;
    if (r.is_black() && t.is_black()) m.none = true;   // `return` before a BSDF is made (translucent.rs:72-74): every hit is skipped like Material "none" — until a texture replaces reflect / transmit
    // each lobe remembers the reflect / transmit factor of its product, so that a Kd / Ks texture can be evaluated per hit (set_material_texture)
    auto feed = [&](int param, int field) { if (m.param_lobe[param] < 0) { m.param_lobe[param] = (int)m.lobes.size(); m.param_field[param] = field; } else { m.param_lobe2[param] = (int)m.lobes.size(); m.param_field2[param] = field; } };
    Spec d = spec_clamp0(spec3(kd));
    if (!d.is_black()) {
        if (!r.is_black()) { Lobe l; l.kind = LK_LAMBERT; l.type = BX_REFL | BX_DIFF; l.r = r * d; l.pre = r; l.pre_raw_test = true; feed(0, 0); m.lobes.push_back(l); }
        if (!t.is_black()) { Lobe l; l.kind = LK_LAMBERT_T; l.type = BX_TRANS | BX_DIFF; l.t = t * d; l.pre = t; l.pre_raw_test = true; feed(0, 1); m.lobes.push_back(l); }
    }
    Spec sp = spec_clamp0(spec3(ks));
    if (!sp.is_black() && (!r.is_black() || !t.is_black())) {
        Float rough = remap ? roughness_to_alpha(roughness) : roughness;
        m.rough_remap = remap != 0;
        if (!r.is_black()) { Lobe l; l.kind = LK_MICRO_R; l.type = BX_REFL | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f; l.r = r * sp; l.pre = r; l.pre_raw_test = true; set_tr(l, rough, rough);
                             feed(1, 0); m.rough_lobe = (int)m.lobes.size(); m.lobes.push_back(l); }
        if (!t.is_black()) { Lobe l; l.kind = LK_MICRO_T; l.type = BX_TRANS | BX_GLOSSY; l.fresnel = FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f; l.t = t * sp; l.pre = t; l.pre_raw_test = true; set_tr(l, rough, rough);
                             feed(1, 1); (m.rough_lobe < 0 ? m.rough_lobe : m.rough_lobe2) = (int)m.lobes.size(); m.lobes.push_back(l); }
    }
    return push_material(s, m, out_id);
}
int oracle_add_material_mix(OracleScene* s, uint32_t m1, uint32_t m2, const float amount[3], uint32_t* out_id) {  // mix.rs:51-88
    if (!s || !amount || m1 >= s->sc.materials.size() || m2 >= s->sc.materials.size()) return -1;
    Material m; m.general = true;
    Spec s1 = spec_clamp0(spec3(amount));
    Spec s2 = spec_clamp0(Spec(1.0f) - s1);
    const Material a = s->sc.materials[m1], b = s->sc.materials[m2];
    if (a.lobes.size() + b.lobes.size() > 8) { s->err = "mix: more than MAX_BXDFS = 8 lobes (bsdf.rs:119-125 asserts)"; return -1; }
    for (Lobe l : a.lobes) { if (l.n_scale >= 2) { s->err = "mix nested deeper than two levels"; return -5; } l.scale[l.n_scale++] = s1; m.lobes.push_back(l); }
    for (Lobe l : b.lobes) { if (l.n_scale >= 2) { s->err = "mix nested deeper than two levels"; return -5; } l.scale[l.n_scale++] = s2; m.lobes.push_back(l); }
    // mix.rs:63-76: m1 bumps `si` itself, m2 a clone of it; the mixture's BSDF is then made on `si` (BSDF::new(&si.hit, &si.shading, None)), so the first
    // material's bump map shapes the frame of every lobe and the second one's changes nothing that is used
    m.bump_tex = a.bump_tex;
    if (a.amount_tex >= 0 || b.amount_tex >= 0) { s->err = "mix of a mix whose amount is a texture: not modelled"; return -5; }
    if (a.opacity_tex >= 0 && b.opacity_tex >= 0) { s->err = "mix of two uber materials with opacity textures: not modelled"; return -5; }
    m.opacity_tex = a.opacity_tex >= 0 ? a.opacity_tex : b.opacity_tex;
    m.made_as = 4; m.mix_n1 = (int)a.lobes.size();
    m.textured = a.textured || b.textured;   // the sub-materials' textures are evaluated per hit, each lobe kept or dropped as its own material would (mix.rs:63-87)
    return push_material(s, m, out_id);
}

// BSDF probe for pinning tests: the material's BSDF in the canonical frame (ns = ng = +z, ss = +x).
//   op 0: out[0..2] = f(wo, wi, flags), out[3] = pdf(wo, wi, flags)
//   op 1: sample_f(wo, u, flags): out[0..2] = f, out[3] = pdf, out[4..6] = wi, out[7] = sampled BxDFType
//   op 2: out[0] = num_components(flags), out[1] = number of lobes, out[2] = bsdf.eta
int oracle_bsdf_probe(OracleScene* s, uint32_t material, int op, const float wo[3], const float wi[3], const float u[2], int flags, float out[8]) {
    if (!s || material >= s->sc.materials.size() || !out) return -1;
    const Material& m = s->sc.materials[material];
    Renderer::BSDF b; b.ns = b.ng = V3(0, 0, 1); b.ss = V3(1, 0, 0); b.ts = cross(b.ns, b.ss);
    b.lobes = m.lobes.data(); b.n = (int)m.lobes.size(); b.eta = m.bsdf_eta;
    for (int i = 0; i < 8; i++) out[i] = 0.0f;
    if (op == 0) {
        Spec f = b.f(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), flags);
        out[0] = f.c[0]; out[1] = f.c[1]; out[2] = f.c[2]; out[3] = b.pdf(V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), flags);
    } else if (op == 1) {
        Spec f; Float pdf; V3 w; int st = 0;
        b.sample_f(V3(wo[0], wo[1], wo[2]), V2(u[0], u[1]), f, pdf, w, flags, &st);
        out[0] = f.c[0]; out[1] = f.c[1]; out[2] = f.c[2]; out[3] = pdf; out[4] = w.x; out[5] = w.y; out[6] = w.z; out[7] = (float)st;
    } else {
        out[0] = (float)b.num_components(flags); out[1] = (float)b.n; out[2] = b.eta;
    }
    return 0;
}

// SpatialLightDistribution of the last render: out[0..2] voxel resolution, out[3] distributions created
// ("SpatialLightDistribution/Distributions created", spatial.rs:17-21).
int oracle_spatial_stats(OracleScene* s, uint64_t out[4]) {
    if (!s || !out) return -1;
    for (int i = 0; i < 3; i++) out[i] = (uint64_t)s->r.n_voxels[i];
    out[3] = s->r.spatial_created;
    return 0;
}
// One voxel's distribution, computed directly (compute_distribution, spatial.rs:90-160): func[n_lights], cdf[n_lights+1].
int oracle_spatial_voxel(OracleScene* s, const int pi[3], float* out_func, float* out_cdf, float* out_func_int) {
    if (!s || !pi || !s->built) return -1;
    Renderer& r = s->r;
    r.spatial_init(64);
    Dist1D d; r.spatial_compute(pi, d);
    for (size_t i = 0; i < d.func.size(); i++) out_func[i] = d.func[i];
    for (size_t i = 0; i < d.cdf.size(); i++) out_cdf[i] = d.cdf[i];
    *out_func_int = d.func_int;
    return 0;
}
int oracle_render_path(OracleScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size,
                       int tile_part, int tile_parts, float* out_xyz, float* out_weight, OracleStats* st) {
    return oracle_render_path_ex(s, max_depth, rr_threshold, light_strategy, pixel_bounds, tile_size, tile_part, tile_parts, out_xyz, out_weight, st, 0, 0, nullptr);
}
// ORACLE ONLY: render_path runs WhittedIntegrator::li instead of PathIntegrator::li (0 restores the path integrator); max_depth keeps its meaning
int oracle_set_integrator(OracleScene* s, int kind) {
    if (!s || kind < 0 || kind > 1) return -1;
    s->r.integrator = kind;
    return 0;
}
int oracle_film_to_rgb(const OracleScene* s, const float* xyz, const float* weight, float* rgb) {
    if (!s || !xyz || !weight || !rgb) return -1;
    s->r.film_to_rgb(xyz, weight, rgb); return 0;
}
int oracle_generate_camera_rays(OracleScene* s, const int pb[4], uint32_t sample_index, OracleRay* out_rays, float* out_pfilm) {
    if (!s || !pb || !out_rays) return -1;
    if (!s->have_camera || !s->have_sampler) return -2;
    size_t k = 0;
    auto emit = [&](auto& sp) {
        for (int y = pb[1]; y < pb[3]; y++)
            for (int x = pb[0]; x < pb[2]; x++, k++) {
                sp.start_pixel(x, y); sp.set_sample_number(sample_index);
                V2 fs = sp.get_2d(); V2 pf((Float)x + fs.x, (Float)y + fs.y); Float tm = sp.get_1d(); V2 pl = sp.get_2d();
                Ray r = s->r.generate_ray(pf, tm, pl);
                OracleRay& o = out_rays[k];
                o.o[0] = r.o.x; o.o[1] = r.o.y; o.o[2] = r.o.z; o.t_max = r.t_max; o.d[0] = r.d.x; o.d[1] = r.d.y; o.d[2] = r.d.z; o.time = r.time;
                if (out_pfilm) { out_pfilm[2 * k] = pf.x; out_pfilm[2 * k + 1] = pf.y; }
            }
    };
    if (s->r.scfg.kind == 0) { HaltonSampler sp(s->r.scfg); emit(sp); }
    else { if (!s->r.scfg.sobol.m32) return -2; SobolSampler sp(s->r.scfg); emit(sp); }
    return 0;
}

void oracle_set_libm_mode(int m) { g_libm_mode = m; }

// ---- oracle-only extras -----------------------------------------------------------------------------------------------
int oracle_record_rays(OracleScene* s, uint64_t cap) { if (!s) return -1; s->rec.cap = cap; s->rec.regular.clear(); s->rec.shadow.clear(); return 0; }
uint64_t oracle_recorded_count(OracleScene* s, int shadow) { return s ? (shadow ? s->rec.shadow.size() : s->rec.regular.size()) : 0; }
int oracle_recorded_rays(OracleScene* s, int shadow, OracleRay* out) {
    if (!s || !out) return -1;
    const std::vector<Ray>& v = shadow ? s->rec.shadow : s->rec.regular;
    for (size_t i = 0; i < v.size(); i++) {
        out[i].o[0] = v[i].o.x; out[i].o[1] = v[i].o.y; out[i].o[2] = v[i].o.z; out[i].t_max = v[i].t_max;
        out[i].d[0] = v[i].d.x; out[i].d[1] = v[i].d.y; out[i].d[2] = v[i].d.z; out[i].time = v[i].time;
    }
    return 0;
}
uint64_t oracle_bvh_node_count(const OracleScene* s) { return s ? s->sc.nodes.size() : 0; }
int oracle_bvh_nodes(const OracleScene* s, void* out32B) { if (!s) return -1; std::memcpy(out32B, s->sc.nodes.data(), s->sc.nodes.size() * 32); return 0; }
int oracle_bvh_ordered_prims(const OracleScene* s, uint32_t* out) { if (!s) return -1; std::memcpy(out, s->sc.ordered_prims.data(), s->sc.ordered_prims.size() * 4); return 0; }

// known-answer probes (tests/test_oracle_kat.py)
void oracle_rng_u32(uint64_t seq, int use_default, uint32_t* out, int n) { RNG r = use_default ? RNG() : RNG(seq); for (int i = 0; i < n; i++) out[i] = r.uniform_u32(); }
void oracle_halton_perm(int prime_index, uint16_t* out) { const auto& t = ld_tables(); std::memcpy(out, &t.perms[t.prime_sums[prime_index]], t.primes[prime_index] * 2); }
uint32_t oracle_prime(int i) { return ld_tables().primes[i]; }
uint32_t oracle_prime_sum(int i) { return ld_tables().prime_sums[i]; }
float oracle_radical_inverse(int base_index, uint64_t a) { return radical_inverse(base_index, a); }
float oracle_scrambled_radical_inverse(int base_index, uint64_t a) { const auto& t = ld_tables(); return scrambled_radical_inverse(base_index, a, &t.perms[t.prime_sums[base_index]]); }
// Halton sample for (pixel, sample number, dimension) using the configured sampler
float oracle_sampler_value(OracleScene* s, int x, int y, uint32_t sample, uint32_t dim) {
    HaltonSampler sp(s->r.scfg); sp.start_pixel(x, y); sp.set_sample_number(sample);
    return sp.sample_dimension(sp.interval_index, dim);
}

// geometry probes that replay the reference's proptests (core/src/geometry/*.rs #[cfg(test)])
// op: 0 dot 1 cross 2 normalize 3 length 4 abs 5 min/max component 6 max_dimension 7 permute(xyz->a[3..6]) 8 ray.at
//     9 coordinate_system 10 matrix inverse (16 in, 16 out) 11 face_forward 12 distance_squared 13 box test
void oracle_geom_op(int op, const float* a, float* out) {
    V3 u(a[0], a[1], a[2]), v(a[3], a[4], a[5]);
    switch (op) {
    case 0: out[0] = dot(u, v); break;
    case 1: { V3 c = cross(u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 2: { V3 c = normalize(u); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 3: out[0] = length(u); out[1] = length_squared(u); break;
    case 4: { V3 c = vabs(u); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 5: out[0] = max_component(u); break;
    case 6: out[0] = (float)max_dimension(u); break;
    case 7: { V3 c = permute(u, (int)a[3], (int)a[4], (int)a[5]); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 8: { V3 c = u + v * a[6]; out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 9: { V3 b, c; coordinate_system(u, b, c); out[0] = b.x; out[1] = b.y; out[2] = b.z; out[3] = c.x; out[4] = c.y; out[5] = c.z; break; }
    case 10: { M4 m = inverse(m4_from(a)); std::memcpy(out, m.m, 64); break; }
    case 11: { V3 c = face_forward(u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 12: out[0] = distance_squared(u, v); break;
    case 13: {  // a: box pmin, pmax, ray o, d, t_max
        Bounds3 b; b.pmin = u; b.pmax = v; Ray r(V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11]), a[12], 0);
        V3 inv(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z); int neg[3] = {inv.x < 0, inv.y < 0, inv.z < 0};
        out[0] = Scene::box_hit(b, r, inv, neg) ? 1.0f : 0.0f; break;
    }
    // the rest of the Vector3 / Point3 / Normal3 proptests (one POD for the three types): a[6] = scalar
    case 14: { V3 c = u + v; out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 15: { V3 c = u - v; out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 16: { V3 c = u * a[6], d = a[6] * u; out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = d.x; out[4] = d.y; out[5] = d.z; break; }
    case 17: { V3 c = u / a[6]; out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 18: { V3 c = -u; out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 19: { V3 c = vmin(u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 20: { V3 c = vmax(u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 21: out[0] = min_component(u); break;
    case 22: { V3 c = vfloor(u); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 23: { V3 c = vceil(u); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 24: { V3 c = lerp(a[6], u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
    case 25: out[0] = distance(u, v); break;
    case 26: out[0] = abs_dot(u, v); break;
    case 27: out[0] = has_nans(u) ? 1.0f : 0.0f; break;
    case 28: {  // Ray::scale_differentials (ray.rs:90-99): a = o, d, s, rx_origin, ry_origin, rx_direction, ry_direction -> the four scaled
        Ray r(u, v, INF, 0.0f); r.has_diff = true;
        r.rx_o = V3(a[7], a[8], a[9]); r.ry_o = V3(a[10], a[11], a[12]); r.rx_d = V3(a[13], a[14], a[15]); r.ry_d = V3(a[16], a[17], a[18]);
        Renderer::scale_differentials(r, a[6]);
        const V3 q[4] = {r.rx_o, r.ry_o, r.rx_d, r.ry_d};
        for (int k = 0; k < 4; k++) { out[3 * k] = q[k].x; out[3 * k + 1] = q[k].y; out[3 * k + 2] = q[k].z; }
        break;
    }
    }
}
// Vector2 / Point2 probes: u = a[0..1], v = a[2..3], scalar a[4], permutation a[5..6]
void oracle_geom2_op(int op, const float* a, float* out) {
    V2 u(a[0], a[1]), v(a[2], a[3]);
    auto put = [&](V2 c) { out[0] = c.x; out[1] = c.y; };
    switch (op) {
    case 0: out[0] = dot(u, v); out[1] = abs_dot(u, v); break;
    case 2: put(normalize(u)); break;
    case 3: out[0] = length(u); out[1] = length_squared(u); break;
    case 4: put(vabs(u)); break;
    case 5: out[0] = max_component(u); out[1] = min_component(u); break;
    case 6: out[0] = (float)max_dimension(u); break;
    case 7: put(permute(u, (int)a[5], (int)a[6])); break;
    case 12: out[0] = distance_squared(u, v); out[1] = distance(u, v); break;
    case 14: put(u + v); break;
    case 15: put(u - v); break;
    case 16: { V2 c = u * a[4], d = a[4] * u; out[0] = c.x; out[1] = c.y; out[2] = d.x; out[3] = d.y; break; }
    case 17: put(u / a[4]); break;
    case 18: put(-u); break;
    case 19: put(vmin(u, v)); break;
    case 20: put(vmax(u, v)); break;
    case 22: put(vfloor(u)); break;
    case 23: put(vceil(u)); break;
    case 24: put(lerp(a[4], u, v)); break;
    case 27: out[0] = has_nans(u) ? 1.0f : 0.0f; break;
    }
}
// Bounds2 probes (bounds2.rs).  b1 = a[0..3] and b2 = a[4..7] are taken as given (x0, y0, x1, y1; no sorting), p = a[8..9], scalar a[10].
// op: 0 new(a[0..1], a[2..3]) 1 empty 2 from(p) 3 is_empty 4 diagonal 5 area 6 maximum_extent 7 overlaps 8 offset 9 contains 10 contains_exclusive
//     11 bounding_circle 12 lerp(t = p) 13 expand 14 corner(k = a[10]) 15 union(p) 16 union(b2) 17 intersect(b2) 18 iterate (Bounds2i: out = n, then the points)
}  // extern "C"
template <class T> static void bounds2_op(int op, const T* a, T* out) {
    typedef B2<T> B;
    const B b1 = B::raw(a[0], a[1], a[2], a[3]), b2 = B::raw(a[4], a[5], a[6], a[7]);
    auto put = [&](const B& b) { out[0] = b.x0; out[1] = b.y0; out[2] = b.x1; out[3] = b.y1; };
    switch (op) {
    case 0: put(B::make(a[0], a[1], a[2], a[3])); break;
    case 1: put(B::empty()); break;
    case 2: put(B::from_point(a[8], a[9])); break;
    case 3: out[0] = b1.is_empty() ? 1 : 0; break;
    case 4: b1.diagonal(out[0], out[1]); break;
    case 5: out[0] = b1.area(); break;
    case 6: out[0] = (T)b1.maximum_extent(); break;
    case 7: out[0] = b1.overlaps(b2) ? 1 : 0; break;
    case 9: out[0] = b1.contains(a[8], a[9]) ? 1 : 0; break;
    case 10: out[0] = b1.contains_exclusive(a[8], a[9]) ? 1 : 0; break;
    case 13: put(b1.expand(a[10])); break;
    case 14: b1.corner((int)a[10], out[0], out[1]); break;
    case 15: put(b1.union_p(a[8], a[9])); break;
    case 16: put(b1.union_b(b2)); break;
    case 17: put(b1.intersect(b2)); break;
    }
}
extern "C" {
void oracle_bounds2f_op(int op, const float* a, float* out) {
    const Bounds2f b1 = Bounds2f::raw(a[0], a[1], a[2], a[3]);
    if (op == 8) b1.offset(a[8], a[9], out[0], out[1]);
    else if (op == 11) { V2 c; Float r; b2_bounding_circle(b1, c, r); out[0] = c.x; out[1] = c.y; out[2] = r; }
    else if (op == 12) { V2 c = b2_lerp(b1, V2(a[8], a[9])); out[0] = c.x; out[1] = c.y; }
    else bounds2_op<float>(op, a, out);
}
void oracle_bounds2i_op(int op, const int* a, int* out, int out_cap) {
    if (op == 18) {
        int n = 0;
        Bounds2i::raw(a[0], a[1], a[2], a[3]).for_each([&](int x, int y) { if (2 * n + 2 < out_cap) { out[1 + 2 * n] = x; out[2 + 2 * n] = y; } n++; });
        out[0] = n;
    } else bounds2_op<int>(op, a, out);
}
// scene-construction helpers restating the reference's Transform factories (transform.rs) for host-side tests
void oracle_look_at(const float pos[3], const float look[3], const float up[3], float out_m[16], float out_minv[16]) {
    Transform t = t_look_at(V3(pos[0], pos[1], pos[2]), V3(look[0], look[1], look[2]), V3(up[0], up[1], up[2]));
    std::memcpy(out_m, t.m.m, 64); std::memcpy(out_minv, t.m_inv.m, 64);
}
// raster_to_camera for a perspective camera (camera.rs:276-306 + perspective_camera.rs:47-66)
void oracle_perspective_raster_to_camera(float fov, int xres, int yres, const float screen[4], float out_m[16]) {
    Transform c2s = t_perspective(fov, 1e-2f, 1000.0f);
    Transform s2r = t_scale((Float)xres, (Float)yres, 1.0f) * t_scale(1.0f / (screen[1] - screen[0]), 1.0f / (screen[2] - screen[3]), 1.0f) *
                    t_translate(V3(-screen[0], -screen[3], 0.0f));
    Transform r2s = s2r.inv();
    Transform r2c = c2s.inv() * r2s;
    std::memcpy(out_m, r2c.m.m, 64);
}
// raster_to_camera for an orthographic camera: Transform::orthographic(0, 1) (orthographic_camera.rs:50-56, transform.rs:222-225)
void oracle_orthographic_raster_to_camera(int xres, int yres, const float screen[4], float out_m[16]) {
    Transform c2s = t_scale(1.0f, 1.0f, 1.0f / (1.0f - 0.0f)) * t_translate(V3(0.0f, 0.0f, -0.0f));
    Transform s2r = t_scale((Float)xres, (Float)yres, 1.0f) * t_scale(1.0f / (screen[1] - screen[0]), 1.0f / (screen[2] - screen[3]), 1.0f) *
                    t_translate(V3(-screen[0], -screen[3], 0.0f));
    Transform r2c = c2s.inv() * s2r.inv();
    std::memcpy(out_m, r2c.m.m, 64);
}
void oracle_transform_compose(int kind, const float* p, float out_m[16], float out_minv[16]) {  // 0 translate 1 scale 2 rotate(deg,axis)
    Transform t;
    if (kind == 0) t = t_translate(V3(p[0], p[1], p[2]));
    else if (kind == 1) t = t_scale(p[0], p[1], p[2]);
    else t = t_rotate_axis(p[0], V3(p[1], p[2], p[3]));
    std::memcpy(out_m, t.m.m, 64); std::memcpy(out_minv, t.m_inv.m, 64);
}

}  // extern "C"
