/*
 * pbrt_hip.h — C ABI of the MI355X-native (gfx950) wavefront path tracer that replaces the hot path of
 * hackmad/pbrt-v3-rs:  SamplerIntegrator::render -> PathIntegrator::li -> BVHAccel::intersect/intersect_p ->
 * Triangle::intersect/intersect_p.
 *
 * Every entry point below names the reference interface it replaces (path:line relative to the reference repo).
 * Plain C types only: pointers + sizes, caller-allocated outputs, int status codes.  No callbacks, no C++ or
 * torch types, no exceptions across the boundary.  A handle is NOT re-entrant (one in-flight call per handle).
 * pbrt_hip_scene_create gives a handle that drives one GPU (one process per GPU: multi-GPU = one handle per rank, the ranks' film
 * tiles gathered by the host); pbrt_hip_scene_create_multi gives one handle that drives several GPUs of a node
 * from one process, the library gathering the film tiles itself over RCCL.
 *
 * Inputs are borrowed for the duration of the call and copied.  All floats are IEEE binary32 ("Float = f32",
 * core/src/pbrt/common.rs:13).  Matrices are row-major float[16], m[r*4+c] (core/src/geometry/matrix4x4.rs:13).
 */
#ifndef PBRT_HIP_H
#define PBRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PbrtHipScene PbrtHipScene;

/* ---- status codes (reference behaviour on the same conditions is panic!/error!) ------------- */
enum {
    PBRT_HIP_OK = 0,
    PBRT_HIP_ERR_INVALID_ARG = -1,  /* NULL handle, bad sizes, out-of-range ids                                  */
    PBRT_HIP_ERR_STATE = -2,        /* call order violated (render before build_accel, no camera, ...)           */
    PBRT_HIP_ERR_DEVICE = -3,       /* a HIP call failed; pbrt_hip_last_error() holds hipGetErrorString()        */
    PBRT_HIP_ERR_NO_DEVICE = -4,    /* no gfx950 device visible: the product path never falls back to the CPU    */
    PBRT_HIP_ERR_UNSUPPORTED = -5,  /* feature outside the hot-path scope (SURVEY §8 "next"/"out of scope")      */
    PBRT_HIP_ERR_OOM = -6
};

/* 32-byte ray = core/src/geometry/ray.rs:10-28 without differentials/medium (dead on this path). */
typedef struct PbrtHipRay {
    float o[3];
    float t_max;
    float d[3];
    float time;
} PbrtHipRay;

/* 32-byte closest-hit record: what Primitive::intersect (core/src/primitive.rs:22) hands back, reduced to the
 * ray-dependent quantities of Triangle::intersect (shapes/src/triangle.rs:438-545): t, barycentrics and the
 * primitive.  prim = index into the concatenation of all meshes' triangles in pbrt_hip_add_mesh order
 * (= position in the reference's `primitives: &[ArcPrimitive]` before BVH reordering); 0xFFFFFFFF on a miss.
 * For a hit inside an object instance prim is the triangle's number in add_mesh order (object meshes included) and pad[1] =
 * instance number + 1 (add_instance call order); t, b0..b2 are those of the instance-space ray, as TransformedPrimitive::intersect
 * leaves them in `r.t_max` (transformed_primitive.rs:56).  pad[1] = 0 otherwise.
 * pad[0] carries the library's leaf-order index of the hit triangle and pad[2] its material class | material id << 3 (internal shortcuts for the shade stage and its work queues); ignore them. */
typedef struct PbrtHipHit {
    float t;
    uint32_t prim;
    float b0, b1, b2;
    uint32_t pad[3];
} PbrtHipHit;

/* Counters with the reference's own names (core/src/scene.rs:13-23, core/src/integrator/sampler_integrator.rs:19-23).
 * Mrays/s := (regular_rays + shadow_rays) / render seconds. */
typedef struct PbrtHipStats {
    uint64_t camera_rays;        /* "Integrator/Camera rays traced"                         */
    uint64_t regular_rays;       /* "Intersections/Regular ray intersection tests"          */
    uint64_t shadow_rays;        /* "Intersections/Shadow ray intersection tests"           */
    uint64_t paths_zero_radiance;/* "Integrator/Zero-radiance paths" numerator (path.rs:168) */
    uint64_t paths_total;        /* its denominator (path.rs:164)                            */
    double render_seconds;       /* device time of the render phase (HIP events)             */
    double extend_seconds;       /* of which: traversal launches that carry closest-hit rays (from the second wavefront round on they also
                                    carry that round's shadow rays: one launch per round) */
    double shadow_seconds;       /* of which: any-hit-only traversal launches (0 unless PBRT_HIP_SPLIT_TRAVERSAL is set) */
    double shade_seconds;        /* of which: raygen + shade + film kernels                  */
    uint64_t extend_launches;    /* number of launches counted in extend_seconds             */
    uint64_t shadow_launches;    /* number of launches counted in shadow_seconds             */
    uint64_t light_distributions_created; /* "SpatialLightDistribution/Distributions created" (light_distrib/spatial.rs:17-21) */
} PbrtHipStats;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
int pbrt_hip_device_count(void);                          /* <0 on HIP failure */
PbrtHipScene* pbrt_hip_scene_create(int device_ordinal);  /* NULL if no usable device */
/* One handle over several devices of this node (scene_create(device_ordinals, n_devices); NULL / 0 = every visible device).  It replaces
 * the reference's tile loop over worker threads (core/src/integrator/sampler_integrator.rs:254-296): capture calls act on the handle, pbrt_hip_render_path
 * replicates the scene into every device's HBM, deals the handle's 16x16 tiles round-robin to the devices (tile enumeration :254-259, :314-336), gathers the
 * per-tile film buffers on the first device — ncclSend / ncclRecv over xGMI, the path's one exchange step; RCCL is loaded when first needed — and merges
 * them there in increasing tile index (Film::merge_film_tile, core/src/film/mod.rs:220-248): the film equals the one-device film bit for bit.
 * An ordinal may repeat (contexts sharing a GPU, device-to-device copies instead of RCCL): a rehearsal aid for boxes with one GPU.
 * The batch and *_tiles_device entry points of such a handle act on its first device only. */
PbrtHipScene* pbrt_hip_scene_create_multi(const int* device_ordinals, int n_devices);
int pbrt_hip_scene_devices(const PbrtHipScene*, int* out_ordinals, int capacity);   /* returns the number of devices behind the handle */
/* Self-test of the RCCL binding on the handle's devices (a one-rank communicator on a single-device handle): every device sends n_floats floats to the
 * first one through the same ncclSend / ncclRecv group the film-tile gather uses; out_wrong = floats that arrived wrong. */
int pbrt_hip_selftest_rccl_gather(PbrtHipScene*, uint32_t n_floats, uint64_t* out_wrong);
void pbrt_hip_scene_destroy(PbrtHipScene*);
const char* pbrt_hip_last_error(const PbrtHipScene*);     /* NUL-terminated, owned by the library; NULL handle -> global */

/* ---- scene capture (replaces the string factories, api/src/graphics_state.rs:254-720) -------------------- */

/* MatteMaterial with constant textures (materials/src/matte.rs:47-92). sigma in degrees, clamped to [0,90]. */
int pbrt_hip_add_material_matte(PbrtHipScene*, const float kd_rgb[3], float sigma_deg, uint32_t* out_id);

/* More materials, all with constant textures: the library builds the BxDF list their compute_scattering_functions would
 * (allow_multiple_lobes = true, TransportMode::Radiance, as PathIntegrator calls it, integrators/src/path.rs:143).  Roughness is
 * remapped by TrowbridgeReitzDistribution::roughness_to_alpha when remap_roughness != 0 (the scene-file default).
 *   mirror   materials/src/mirror.rs:40-62     SpecularReflection(Kr, FresnelNoOp)
 *   plastic  materials/src/plastic.rs:50-82    LambertianReflection(Kd) + MicrofacetReflection(Ks, TR(rough), Dielectric(1.5, 1))
 *   glass    materials/src/glass.rs:62-118     FresnelSpecular(Kr, Kt, 1, eta) if both roughnesses are 0, else MicrofacetReflection +
 *                                              MicrofacetTransmission over TR(urough, vrough)
 *   metal    materials/src/metal.rs:62-98      MicrofacetReflection(1, TR(urough, vrough), Conductor(1, eta, k)); eta/k as RGB (the
 *                                              reference's copper default needs its spectral tables: pass them explicitly)
 *   uber     materials/src/uber.rs:116-186     opacity pass-through + Lambert(Kd) + Microfacet(Ks) + SpecularReflection(Kr) +
 *                                              SpecularTransmission(Kt); pass urough = vrough = roughness when the file gives one value
 *   substrate   materials/src/substrate.rs:55-84   FresnelBlend(Kd, Ks, TR(urough, vrough))
 *   translucent materials/src/translucent.rs:57-112 Lambertian R/T (reflect*Kd, transmit*Kd) + Microfacet R/T (reflect*Ks, transmit*Ks), eta 1.5;
 *                                               reflect = transmit = 0: no BSDF at any hit (the reference returns before making one), i.e. Material "none"
 *   mix         materials/src/mix.rs:51-88        the two materials' lobes as ScaledBxDF(amount) / ScaledBxDF(1 - amount); at most 8 lobes,
 *                                               mixes of mixes up to two levels
 * Bump maps are outside the scope (constant textures have no gradient: Material::bump is the identity for them). */
int pbrt_hip_add_material_none(PbrtHipScene*, uint32_t* out_id);   /* Material "none" / "": no BSDF; PathIntegrator::li passes through the surface without
                                                                       counting a bounce (integrators/src/path.rs:142-150) */
int pbrt_hip_add_material_mirror(PbrtHipScene*, const float kr_rgb[3], uint32_t* out_id);
int pbrt_hip_add_material_plastic(PbrtHipScene*, const float kd_rgb[3], const float ks_rgb[3], float roughness, int remap_roughness, uint32_t* out_id);
int pbrt_hip_add_material_glass(PbrtHipScene*, const float kr_rgb[3], const float kt_rgb[3], float uroughness, float vroughness, float eta,
                                int remap_roughness, uint32_t* out_id);
int pbrt_hip_add_material_metal(PbrtHipScene*, const float eta_rgb[3], const float k_rgb[3], float uroughness, float vroughness, int remap_roughness,
                                uint32_t* out_id);
int pbrt_hip_add_material_uber(PbrtHipScene*, const float kd_rgb[3], const float ks_rgb[3], const float kr_rgb[3], const float kt_rgb[3],
                               const float opacity_rgb[3], float uroughness, float vroughness, float eta, int remap_roughness, uint32_t* out_id);

int pbrt_hip_add_material_substrate(PbrtHipScene*, const float kd_rgb[3], const float ks_rgb[3], float uroughness, float vroughness, int remap_roughness,
                                    uint32_t* out_id);
int pbrt_hip_add_material_translucent(PbrtHipScene*, const float kd_rgb[3], const float ks_rgb[3], const float reflect_rgb[3], const float transmit_rgb[3],
                                      float roughness, int remap_roughness, uint32_t* out_id);
int pbrt_hip_add_material_mix(PbrtHipScene*, uint32_t material1, uint32_t material2, const float amount_rgb[3], uint32_t* out_id);

/* TriangleMesh (shapes/src/triangle.rs:75-113): P/N/S must ALREADY be in world space exactly as TriangleMesh::new
 * leaves them (:93-99).  N, S, UV may be NULL.  One GeometricPrimitive per triangle (api/src/lib.rs:783-812).
 * first_area_light_id: -1, or the id returned by pbrt_hip_add_light_diffuse_area for the SAME n_tris (triangle k
 * is bound to light id+k).  flags: bit0 reverse_orientation, bit1 transform_swaps_handedness
 * (core/src/geometry/shape.rs:163-177).  alpha / shadowalpha: constant float textures only (triangle.rs:291-312);
 * pass 1.0f,1.0f for the defaults. */
int pbrt_hip_add_mesh(PbrtHipScene*, const float* P, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris,
                      const float* N, const float* S, const float* UV, uint32_t material_id,
                      int32_t first_area_light_id, uint32_t flags, float alpha, float shadow_alpha);

/* Object instancing (api/src/lib.rs:911-1000, core/src/primitives/transformed_primitive.rs:33-73).  Meshes added between
 * object_begin and object_end belong to the object, not to the scene (ObjectBegin/ObjectEnd); add_instance places one
 * TransformedPrimitive(object aggregate, instance_to_world) in the scene's primitive list (ObjectInstance) — nothing if the
 * object is empty.  The object's aggregate uses the split method of pbrt_hip_build_accel; an object with exactly one
 * primitive is used directly (lib.rs:953-971).  Area lights of meshes inside an object definition behave as in the reference
 * ("Area lights not supported with object instancing", lib.rs:877-881): the call succeeds, pbrt_hip_last_error holds that warning, the triangles keep their
 * emission where a path looks at them (SurfaceInteraction::le) and the lights leave the scene's light list — nothing samples them; they must be the lights
 * created last, so that no other light's number changes.  Transforms are static (AnimatedTransform with equal end points). */
int pbrt_hip_object_begin(PbrtHipScene*, uint32_t* out_object_id);
int pbrt_hip_object_end(PbrtHipScene*);
int pbrt_hip_add_instance(PbrtHipScene*, uint32_t object_id, const float instance_to_world[16], const float world_to_instance[16]);

/* Alpha masks: the float textures behind a mesh's `alpha` / `shadowalpha` parameters (TriangleMesh::alpha_mask, shadow_alpha_mask: shapes/src/triangle.rs:291-312),
 * for the mesh added last; 0xFFFFFFFF keeps the constant given to add_mesh.  A candidate hit is rejected where the texture evaluates to exactly 0 at the hit's
 * uv / point, with no differentials (triangle.rs:587-607; intersect_p also consults shadowalpha, :868-898). */
int pbrt_hip_set_last_mesh_alpha_textures(PbrtHipScene*, uint32_t alpha_texture, uint32_t shadow_alpha_texture);

/* ---- textures (textures/src/{constant,scale,mix,imagemap}.rs, core/src/mipmap/mod.rs) -------------------------------
 * Float- and spectrum-valued textures share one id space; a float texture is a spectrum texture whose channels are equal.
 * add_mipmap is generate_mipmap + MIPMap::new (core/src/mipmap/cache.rs:74-120, mod.rs:115-189): `rgb` is width*height
 * texels as the reference's read_image returns them (row 0 = top of the image); the library flips them, applies
 * `scale * (gamma ? inv_gamma_correct(x) : x)` (as_float: to the texel's luminance, ImageTexture<Float>), resamples to powers of two and
 * builds the pyramid.  filtering: 0 trilinear, 1 EWA.  wrap: 0 repeat, 1 black, 2 clamp.  Image file decoding is the host's job.
 * imagemap uses UVMapping2D (su, sv, du, dv) (core/src/texture/mapping/uv_2d.rs); other mappings are not provided yet.
 * mix: (1 - amount) * tex1 + amount * tex2 with `amount` a float texture.  Trees that need more than 6 live values are refused. */
int pbrt_hip_add_mipmap(PbrtHipScene*, int width, int height, const float* rgb, int as_float, float scale, int gamma, int filtering, int wrap,
                        float max_anisotropy, uint32_t* out_mipmap);
int pbrt_hip_add_texture_constant(PbrtHipScene*, const float value[3], uint32_t* out_texture);
int pbrt_hip_add_texture_scale(PbrtHipScene*, uint32_t tex1, uint32_t tex2, uint32_t* out_texture);
int pbrt_hip_add_texture_mix(PbrtHipScene*, uint32_t tex1, uint32_t tex2, uint32_t amount, uint32_t* out_texture);
int pbrt_hip_add_texture_imagemap(PbrtHipScene*, uint32_t mipmap, float su, float sv, float du, float dv, uint32_t* out_texture);
/* Procedural 2D textures over the same uv mapping: CheckerboardTexture2D (textures/src/checkerboard_2d.rs; aa_mode 0 "none", 1 "closedform"),
 * UVTexture (uv.rs), BilerpTexture (bilerp.rs; four constant corner values), DotsTexture (dots.rs; Perlin noise of core/src/texture/common.rs).
 * add_texture_dots takes DotsTexture's fields: `inside` is evaluated inside a dot.  A caller that translates SCENE-FILE parameters must swap them: the reference builds the
 * texture with DotsTexture::new(outside_dot = "inside" parameter, inside_dot = "outside" parameter) (dots.rs:61-66, quirk B13; pbrt_hip_render does the swap). */
int pbrt_hip_add_texture_checkerboard(PbrtHipScene*, uint32_t tex1, uint32_t tex2, float su, float sv, float du, float dv, int aa_mode, uint32_t* out_texture);
int pbrt_hip_add_texture_uv(PbrtHipScene*, float su, float sv, float du, float dv, uint32_t* out_texture);
int pbrt_hip_add_texture_bilerp(PbrtHipScene*, const float v00[3], const float v01[3], const float v10[3], const float v11[3], float su, float sv, float du, float dv,
                                uint32_t* out_texture);
int pbrt_hip_add_texture_dots(PbrtHipScene*, uint32_t inside, uint32_t outside, float su, float sv, float du, float dv, uint32_t* out_texture);
/* The other TextureMapping2D kinds for a 2D texture (imagemap, checkerboard, uv, bilerp, dots; core/src/texture/mapping/): kind 1 SphericalMapping2D and
 * 2 CylindericalMapping2D take params = the row-major 4x4 world_to_texture matrix (the reference passes tex2world.inverse(), textures/src/lib.rs:54-55),
 * 3 PlanarMapping2D takes params = v1[3], v2[3], udelta, vdelta.  Set it before the texture is used as an operand of another texture. */
int pbrt_hip_set_texture_mapping(PbrtHipScene*, uint32_t texture, int kind, const float* params);
/* 3D procedural textures over IdentityMapping3D (core/src/texture/mapping/identity_3d.rs): `m` is the row-major 4x4 matrix the mapping applies to the
 * hit point and to dp/dx, dp/dy — the reference hands it the Texture directive's CTM (tex2world, textures/src/fbm.rs:65), so pass that.
 * FBmTexture, WrinkledTexture (fbm.rs, wrinkled.rs; omega = "roughness", octaves), WindyTexture (windy.rs), MarbleTexture (marble.rs; scale, variation),
 * CheckerboardTexture3D (checkerboard_3d.rs).  fbm / wrinkled / windy are float-valued (three equal channels). */
int pbrt_hip_add_texture_fbm(PbrtHipScene*, const float m[16], float omega, int octaves, uint32_t* out_texture);
int pbrt_hip_add_texture_wrinkled(PbrtHipScene*, const float m[16], float omega, int octaves, uint32_t* out_texture);
int pbrt_hip_add_texture_windy(PbrtHipScene*, const float m[16], uint32_t* out_texture);
int pbrt_hip_add_texture_marble(PbrtHipScene*, const float m[16], float omega, int octaves, float scale, float variation, uint32_t* out_texture);
int pbrt_hip_add_texture_checkerboard3d(PbrtHipScene*, uint32_t tex1, uint32_t tex2, const float m[16], uint32_t* out_texture);
/* Replaces a colour parameter of an existing material by a texture evaluated at every hit (`self.kd.evaluate(..).clamp_default()` in
 * compute_scattering_functions: materials/src/matte.rs:63, plastic.rs:62-70, mirror.rs:53-55, substrate.rs:60-62), with the ray
 * differentials of camera rays (SurfaceInteraction::compute_differentials) driving the MIPMap filter.  Which lobes a hit gets follows the
 * reference's `is_black` tests on the value at that hit.  Texturable so far: matte Kd, plastic Kd / Ks, mirror Kr, substrate Kd / Ks, glass Kr / Kt (glass.rs:72-78)
 * uber Kd / Ks / Kr / Kt (multiplied by its constant opacity, uber.rs:133-160) and translucent Kd / Ks (each feeds a reflection and a transmission
 * lobe; the texel is tested for black BEFORE its product with the constant reflect / transmit, translucent.rs:77-84, :87);
 * the material must have been created with a non-black constant for that parameter.  Scalar parameters: see set_material_float_texture.
 * Materials with per-hit textures may be children of a mix (each lobe is kept or dropped as its own material would, mix.rs:63-87) as long as the
 * mix has at most 6 textured colours and one textured roughness / sigma. */
/* More per-hit parameters (same call): OPACITY — UberMaterial's opacity (uber.rs:126-160: it decides the pass-through lobe, multiplies Kd / Ks / Kr / Kt and picks BSDF::eta);
 * AMOUNT — MixMaterial's amount (mix.rs:59-60: s1 = amount(hit), s2 = 1 - s1, clamped); ETA / K — MetalMaterial's conductor indices (metal.rs:121-125, not clamped).
 * GlassMaterial's u / v roughness go through set_material_float_texture: a hit where both evaluate to 0 gets FresnelSpecular, any other the microfacet pair (glass.rs:110-141).
 * A bump-mapped material may be a child of a mix: the FIRST child's bump map shapes the mixture's shading frame, the second child's has no effect (mix.rs:63-76 builds the
 * BSDF on the interaction the first child bumped). */
/* REFLECT / TRANSMIT — TranslucentMaterial's `reflect` / `transmit` (translucent.rs:70-98): evaluated and clamped at every hit; a lobe is added where its reflect-or-transmit value and its
 * Kd-or-Ks value are both non-black, with their product as colour; where reflect and transmit are BOTH black the reference makes no BSDF for the hit (:72-74) and PathIntegrator::li passes
 * through the surface without counting a bounce (path.rs:142-150) — so does the library, per hit.  (Constants reflect = transmit = 0 behave like Material "none" everywhere.) */
enum { PBRT_HIP_PARAM_KD = 0, PBRT_HIP_PARAM_KS = 1, PBRT_HIP_PARAM_KR = 2, PBRT_HIP_PARAM_KT = 3, PBRT_HIP_PARAM_OPACITY = 4, PBRT_HIP_PARAM_AMOUNT = 5, PBRT_HIP_PARAM_ETA = 6,
       PBRT_HIP_PARAM_K = 7, PBRT_HIP_PARAM_REFLECT = 8, PBRT_HIP_PARAM_TRANSMIT = 9 };
int pbrt_hip_set_material_texture(PbrtHipScene*, uint32_t material, int param, uint32_t texture);
/* Scalar parameters as float textures, evaluated at every hit: fparam 0 = MatteMaterial's sigma (matte.rs:64-70: Lambert where it evaluates to 0, Oren-Nayar elsewhere),
 * 1 / 2 = u / v roughness of the Trowbridge-Reitz distribution of plastic, uber, substrate, translucent and metal (remapped per hit if the material was created with remap_roughness;
 * plastic's and translucent's single `roughness`: set both), and of glass, whose lobe structure switches per hit on `urough == 0 && vrough == 0` (glass.rs:110-141);
 * 3 = `index` of GlassMaterial and UberMaterial (glass.rs:102, uber.rs:128): the hit's index of refraction, taken as the texture gives it, for FresnelSpecular, the
 * FresnelDielectric(1, index) of the reflection lobes, the transmission lobes and — uber — BSDF::eta; an uber's constant opacity then becomes a per-hit constant as well.
 */
int pbrt_hip_set_material_float_texture(PbrtHipScene*, uint32_t material, int fparam, uint32_t texture);
/* Bump mapping: Material::bump (core/src/material.rs:62-101) with the float texture `texture` as displacement, run before the BSDF of a hit is made
 * (every material's `bumpmap` parameter).  Not for Material "none".  Set it before the material becomes a child of a mix. */
int pbrt_hip_set_material_bump(PbrtHipScene*, uint32_t material, uint32_t texture);
/* = add_material_matte((1,1,1), sigma) + set_material_texture(KD) */
int pbrt_hip_add_material_matte_tex(PbrtHipScene*, uint32_t kd_texture, float sigma_degrees, uint32_t* out_material);
/* Test aids: evaluate a texture on the device at explicit contexts (u, v, du/dx, dv/dx, du/dy, dv/dy, p[3], dp/dx[3], dp/dy[3]); read back the pyramid the host built. */
int pbrt_hip_texture_eval_batch(PbrtHipScene*, uint32_t texture, uint64_t n, const float* contexts /*15 per point*/, float* out_rgb /*3 per point*/);
/* The same through the evaluator the renderer uses for every ray but a camera ray (no differentials: an image map's look-up ends in MIPMap::triangle(0, st), mipmap/mod.rs:222-231,
 * :250-262): the contexts' derivative fields are ignored as if zero.  Must equal pbrt_hip_texture_eval_batch on contexts whose derivatives ARE zero. */
int pbrt_hip_texture_eval_batch_nodiff(PbrtHipScene*, uint32_t texture, uint64_t n, const float* contexts /*15 per point*/, float* out_rgb /*3 per point*/);
int pbrt_hip_mipmap_levels(PbrtHipScene*, uint32_t mipmap, int* out_levels, int* out_width_height /*2 per level, <= 16 levels*/);
int pbrt_hip_mipmap_level_texels(PbrtHipScene*, uint32_t mipmap, int level, float* out_rgb);


/* Lights are numbered in call order = position in Scene::lights (core/src/scene.rs:50-75). */
int pbrt_hip_add_light_infinite(PbrtHipScene*, const float L_rgb[3], const float light_to_world[16],
                                const float world_to_light[16]);       /* lights/src/infinite.rs:63-107, constant L */
/* InfiniteAreaLight with a radiance map (`mapname`; lights/src/infinite.rs:52-100): `rgb` = width*height texels as read_image returns them (row 0 = top);
 * the library multiplies them by L, builds the MIPMap (EWA, repeat; no y flip on this path, as in the reference) and the Distribution2D of the 2w x 2h
 * scalar image (compute_scalar_image, :326-369).  Le / sample_li / pdf_li / power then follow :127-211. */
int pbrt_hip_add_light_infinite_map(PbrtHipScene*, const float L[3], int width, int height, const float* rgb, const float light_to_world[16], const float world_to_light[16]);
/* ProjectionLight (lights/src/projection.rs:55-127, :145-203) and GonioPhotometricLight (goniometric.rs:34-126): point lights whose intensity is scaled by an
 * image looked up by direction (MIPMap::new(.., Ewa, Repeat, 8.0), lookup_triangle(st, 0)).  rgb = width x height x 3 floats, top row first, or NULL for "no image"
 * (white inside the frustum / in every direction); I = intensity * scale; light_to_world / world_to_light as for the spot light. */
int pbrt_hip_add_light_projection(PbrtHipScene*, const float I_rgb[3], const float light_to_world[16], const float world_to_light[16], float fov_deg,
                                  int width, int height, const float* rgb);
int pbrt_hip_add_light_goniometric(PbrtHipScene*, const float I_rgb[3], const float light_to_world[16], const float world_to_light[16],
                                   int width, int height, const float* rgb);
int pbrt_hip_add_light_distant(PbrtHipScene*, const float L_rgb[3], const float w_light_world[3]); /* distant.rs:36-50: already transformed+normalized */
int pbrt_hip_add_light_point(PbrtHipScene*, const float I_rgb[3], const float p_world[3]);         /* point.rs:36-55 */
int pbrt_hip_add_light_spot(PbrtHipScene*, const float I_rgb[3], const float light_to_world[16], const float world_to_light[16],
                            float cos_total_width, float cos_falloff_start);                       /* lights/src/spot.rs:27-90 */
int pbrt_hip_add_light_diffuse_area(PbrtHipScene*, const float L_rgb[3], int two_sided, uint32_t n_tris,
                                    uint32_t* out_first_id);           /* lights/src/diffuse.rs:46-82, one per triangle */

/* PerspectiveCamera (cameras/src/perspective_camera.rs:47-95): the two transforms the ray generator uses. */
int pbrt_hip_set_camera_perspective(PbrtHipScene*, const float raster_to_camera[16], const float camera_to_world[16],
                                    float lens_radius, float focal_distance, float shutter_open, float shutter_close);
/* OrthographicCamera (cameras/src/orthographic_camera.rs:39-72, :121-178): same two transforms (raster_to_camera from Transform::orthographic(0, 1):
 * pbrt_hip_host_orthographic_raster_to_camera); rays leave the film point along camera +z, the lens model and the ray differentials are the reference's. */
int pbrt_hip_set_camera_orthographic(PbrtHipScene*, const float raster_to_camera[16], const float camera_to_world[16],
                                     float lens_radius, float focal_distance, float shutter_open, float shutter_close);
/* EnvironmentCamera (cameras/src/environment_camera.rs:27-78): directions over the whole sphere from the film position relative to the film's FULL
 * resolution (pass the same xres / yres as to pbrt_hip_set_film); ray differentials are the Camera trait's finite differences (core/src/camera.rs:29-78). */
int pbrt_hip_set_camera_environment(PbrtHipScene*, const float camera_to_world[16], int xres, int yres, float shutter_open, float shutter_close);

/* Film (core/src/film/mod.rs:89-146).  cropped_pixel_bounds = {x0,y0,x1,y1}.  filter_table = the 16x16 table of
 * Film::new (:117-129).  max_sample_luminance: INFINITY for none. */
int pbrt_hip_set_film(PbrtHipScene*, int xres, int yres, const int cropped_pixel_bounds[4],
                      const float filter_radius[2], const float filter_table_16x16[256], float scale,
                      float max_sample_luminance);

/* Sampler (samplers/src/halton.rs:61-100, sobol.rs:35-63). kind: 0 halton, 1 sobol. sample_bounds = Film::get_sample_bounds. */
int pbrt_hip_set_sampler(PbrtHipScene*, int kind, uint32_t samples_per_pixel, const int sample_bounds[4],
                         int sample_at_pixel_center);

/* Optional: Sobol generator matrices (core/src/sobol_matrices.rs) as data: 1024*52 u32, then VdC 25x? tables.
 * Required before rendering with kind==1; the library does not embed them. */
int pbrt_hip_set_sobol_tables(PbrtHipScene*, const uint32_t* sobol_matrices32, size_t n32,
                              const uint64_t* vdc_matrices, const uint64_t* vdc_matrices_inv, size_t n_vdc_each);

/* BVHAccel::from (accelerators/src/bvh/mod.rs:339-360) = Integrator::preprocess time.  split_method: 0 SAH (sah.rs),
 * 1 HLBVH (hlbvh.rs, the reference's tree incl. its Morton-code quirk), 3 EqualCounts; 2 (Middle) panics in the reference
 * and returns UNSUPPORTED. */
int pbrt_hip_build_accel(PbrtHipScene*, int split_method, int max_prims_in_node);

/* The same with the tree constructed on the GPU: identical topology, leaf order and boxes, so hits do not depend on where the tree was built.
 * split_method 0 (SAH, the reference's default; accelerators/src/bvh/sah.rs:26-367): one level of the tree per round of kernels — bucket boxes by atomics,
 * the reference's cost loop per node, itertools::partition's element order from a prefix sum.  split_method 1 (HLBVH; hlbvh.rs:33-449, morton.rs:33-120):
 * Morton codes, radix sort, the treelets emitted side by side one tree LEVEL per launch, SAH over the treelet roots on the host.  EqualCounts (3) is a host build: UNSUPPORTED here.  Scenes with object
 * instances (SAH only): the scene's aggregate and every instanced object's are built side by side as one forest, level by level. */
int pbrt_hip_build_accel_device(PbrtHipScene*, int split_method, int max_prims_in_node);

/* World bound of the built aggregate (BVHAccel::world_bound, bvh/mod.rs:161-167): {pmin[3], pmax[3]}. */
int pbrt_hip_world_bound(const PbrtHipScene*, float out_bounds[6]);

/* Measurement aid (not a reference interface): the built structure in the device layout.  out[0] interior nodes (64 B each: both children's boxes),
 * out[1] leaf records (48 B each), out[2] / out[3] their bytes, out[4] leaves, out[5] depth, out[6] largest leaf, out[7] build time in microseconds. */
int pbrt_hip_accel_stats(const PbrtHipScene*, uint64_t out[8]);
/* Test aid: copies the structure out (out[0] x 64-byte nodes, out[1] x 48-byte leaf records), from wherever it was built. */
int pbrt_hip_accel_copy(PbrtHipScene*, void* out_nodes, uint64_t node_capacity, void* out_leaf_records, uint64_t record_capacity);

/* ---- the hot path ---------------------------------------------------------------------------------------- */

/* Primitive::intersect on a batch (accelerators/src/bvh/mod.rs:173-226): host buffers, synchronous. */
int pbrt_hip_intersect_batch(PbrtHipScene*, const PbrtHipRay* rays, PbrtHipHit* hits, uint64_t n);
/* Primitive::intersect_p on a batch (bvh/mod.rs:231-283): out[i] = 1 if occluded. */
int pbrt_hip_occluded_batch(PbrtHipScene*, const PbrtHipRay* rays, uint8_t* out_occluded, uint64_t n);

/* Same, device-resident buffers on the handle's device; enqueued on the handle's stream and synchronised
 * before returning unless sync==0.  kernel_ms (may be NULL) receives the HIP-event duration of the kernel alone. */
int pbrt_hip_intersect_batch_device(PbrtHipScene*, const void* d_rays, void* d_hits, uint64_t n, float* kernel_ms);
int pbrt_hip_occluded_batch_device(PbrtHipScene*, const void* d_rays, void* d_occluded, uint64_t n, float* kernel_ms);

/* Measurement aid (not a reference interface): with counting on, traversal launches also tally their work.
 * out[0..2] closest-hit {interior nodes whose box test passed, triangle tests, rays}, out[3..5] the same for any-hit.
 * Reference-format node visits of the closest-hit rays = out[2] + 2*out[0]; of the any-hit rays (which stop early) = out[6],
 * counted as they happen.  out[7] is spare.  get resets the tallies.  Never timed. */
int pbrt_hip_set_traversal_counting(PbrtHipScene*, int on);
int pbrt_hip_get_traversal_counts(PbrtHipScene*, uint64_t out[8]);

/* Integrator::render for PathIntegrator (core/src/integrator/sampler_integrator.rs:243-415 +
 * integrators/src/path.rs:103-284).  Renders the 16x16 (tile_size) sample tiles whose index t satisfies
 * t % tile_parts == tile_part (tile enumeration = sampler_integrator.rs:254-259,314-336), i.e. the whole frame for
 * (0,1).  light_strategy: 0 uniform, 1 power, 2 spatial (the reference's default, core/src/light_distrib/spatial.rs;
 * with one light the reference itself forces uniform, light_distrib/mod.rs:59-64).
 *
 * Output = the per-pixel state Film keeps (core/src/film/mod.rs:33-47): out_xyz[3*i..] = sum of rgb_to_xyz(tile
 * contrib), out_weight[i] = filter weight sum, i indexing cropped_pixel_bounds row-major.  Film::write_image's
 * normalisation (:356-417) is pbrt_hip_film_to_rgb. */
int pbrt_hip_render_path(PbrtHipScene*, int max_depth, float rr_threshold, int light_strategy,
                         const int pixel_bounds[4], int tile_size, int tile_part, int tile_parts,
                         float* out_xyz, float* out_weight, PbrtHipStats* out_stats);

/* Multi-GPU form: writes this rank's FilmTiles (contrib rgb + weight, 4 floats per tile pixel, tiles in
 * increasing index order, each tile's pixel bounds as Film::get_film_tile computes them) into a DEVICE buffer
 * the caller owns (e.g. a torch tensor) so the host can gather them with RCCL; then any rank calls
 * pbrt_hip_merge_tiles on the gathered buffers.  Query sizes with pbrt_hip_tile_buffer_floats. */
int pbrt_hip_tile_buffer_floats(PbrtHipScene*, int tile_size, int tile_part, int tile_parts, uint64_t* out_floats);
int pbrt_hip_render_path_tiles_device(PbrtHipScene*, int max_depth, float rr_threshold, int light_strategy,
                                      const int pixel_bounds[4], int tile_size, int tile_part, int tile_parts,
                                      void* d_tile_buffer, PbrtHipStats* out_stats);
/* Film::merge_film_tile (core/src/film/mod.rs:220-279) in increasing tile order over tile_parts device buffers. */
int pbrt_hip_merge_tiles_device(PbrtHipScene*, int tile_size, int tile_parts, const void* const* d_tile_buffers,
                                float* out_xyz, float* out_weight);

/* Film::get_pixel_rgb (core/src/film/mod.rs:392-417) on the host: rgb[3*i..]. */
int pbrt_hip_film_to_rgb(const PbrtHipScene*, const float* xyz, const float* weight, float* out_rgb);

/* K1 alone, for parity tests of a-S/a-C: camera rays for sample index s of every pixel of pixel_bounds,
 * row-major (get_camera_sample + generate_ray_differential, sampler/mod.rs:45-53, perspective_camera.rs:144-204). */
int pbrt_hip_generate_camera_rays(PbrtHipScene*, const int pixel_bounds[4], uint32_t sample_index,
                                  PbrtHipRay* out_rays, float* out_pfilm_xy);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_HIP_H */
