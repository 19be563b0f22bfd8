// Device-resident scene layout shared by host code and gfx950 kernels (plain PODs, no pointers to host memory).
// Layout rationale is in DESIGN.md §3; reference counterparts are cited per struct.
#pragma once
#include <stdint.h>

// ---- BVH ---------------------------------------------------------------------------------------------------------
// The reference's 32-B LinearBVHNode (accelerators/src/bvh/common.rs:163-179) holds ONE box and is fetched, tested and
// often rejected.  Here an interior node carries BOTH children's boxes (same numbers, same topology), so a rejected child
// is never fetched: one 64-B line per interior visit, 4 x dwordx4 per lane.
// Child reference: bit 31 set -> leaf, low 31 bits = offset of its first TriRec; else index of an interior Node64.
#define PH_LEAF_BIT 0x80000000u
#define PH_INVALID_REF 0xFFFFFFFFu
#define PH_NEED_POP 0xFFFFFFFEu   // traversal-kernel state only (traverse.h): the lane's next reference is on its stack
// Scenes with object instances: a leaf reference whose FIRST record is an instance carries this hint (set by patch_leaf_refs_kernel when the scene is uploaded, api.hip), and a record whose
// successor in the same leaf is an instance carries PH_TRI_NEXT_INST: the traversal kernel then fetches the instance's transform (DeviceScene::inst_extra, addressed by the RECORD's own
// position) together with the record instead of one dependent round trip later.  Only hints: a record met without one is handled the slow way, with the same result.
#define PH_LEAF_INST_HINT 0x40000000u
struct alignas(64) Node64 {
    // planes interleaved so that the dir_is_neg select touches one float2 per axis:
    float x0[2], y0[2], z0[2];  // child 0: {min,max} per axis
    float x1[2], y1[2], z1[2];  // child 1
    uint32_t c0, c1;            // child references
    uint32_t axis;              // split axis of THIS node (bvh/mod.rs:206: visit order = dir_is_neg[axis])
    uint32_t pad;
};

// One triangle in BVH (leaf-contiguous) order: 48 B = the 3 indices + 3 positions a leaf test reads in the reference
// (SURVEY §8d "48·n_t"), pre-gathered so the leaf loop has no index indirection.
#define PH_TRI_LAST 1u   // last triangle of its leaf
#define PH_TRI_BOGUS 2u  // degenerate: Triangle::intersect returns None after the t test (shapes/src/triangle.rs:567-570)
#define PH_TRI_ALPHA0 4u   // mesh alpha texture == 0.0 (triangle.rs:603)
#define PH_TRI_SALPHA0 8u  // mesh shadowalpha texture == 0.0 (triangle.rs:891)
#define PH_TRI_CLASS_SHIFT 8  // bits 8..10: material class of the triangle (shade-side sorting key; 7 = Material "none"), set at build time
#define PH_TRI_MAT_SHIFT 11    // bits 11..22: the triangle's material id (0xFFF: "look it up through the mesh" — a scene with more than 4094 materials), set at build time with the class
#define PH_TRI_KEY_MASK 0x7FFFu  // (flags >> PH_TRI_CLASS_SHIFT) & this = class | material << 3: what the traversal kernel reports in HitOut::pad[2] (shade-side queues, matsort.h)
#define PH_TRI_ALPHATEX 32u  // the mesh's alpha or shadowalpha is a texture: the traversal kernel (ALPHA variants) evaluates it at the candidate hit
#define PH_TRI_NEXT_INST 64u  // the next record of this leaf is an instance (see PH_LEAF_INST_HINT)
#define PH_TRI_INSTANCE 16u  // not a triangle: a TransformedPrimitive (object instance); `prim` = index into DeviceScene::instances
struct alignas(16) TriRec {
    float p0[3]; uint32_t prim;   // prim = index in add_mesh order
    float p1[3]; uint32_t flags;
    float p2[3]; uint32_t mesh;   // mesh id (shade-side shortcut)
};

// One ObjectInstance = TransformedPrimitive (core/src/primitives/transformed_primitive.rs): the object's aggregate lives in the
// same node / TriRec arrays as the scene's; a ray entering it is carried to instance space by transform_ray (transform.rs:451-476).
#define PH_INST_SINGLE 1u    // the object holds exactly one primitive: it is used directly, no aggregate and no root box test (lib.rs:953-971)
#define PH_INST_GENERAL 4u   // (only in an instance's LEAF RECORD, see below) the last row of world-to-instance is not (0, 0, 0, 1): the kernel reads the whole matrix from the InstRec
#define PH_INST_IDENTITY 2u  // instance_to_world is the identity: transform_surface_interaction is skipped (transformed_primitive.rs:58)
// An instance's leaf record (TriRec with PH_TRI_INSTANCE) is filled in when the scene is uploaded: p0 / p1 = the object's root bounds, p2[0] / p2[1] = its root reference and PH_INST_* flags
// (as bit patterns); rows 0 .. 2 of world-to-instance go to DeviceScene::inst_extra[3 * record position ..].  What a ray entering the instance needs then arrives with the record.
struct InstRec {
    float w2i[16], i2w[16];   // row-major 4x4
    float lo[3], hi[3];       // object aggregate's root bounds (instance space)
    uint32_t root_ref, flags;
};

// ---- shading-side geometry (indexed by prim, add_mesh order) ---------------------------------------------------------
struct MeshRec {
    uint32_t vert_base, tri_base, n_tris;
    uint32_t flags;        // bit0 has N, bit1 has S, bit2 has UV, bit3 reverse_orientation, bit4 swaps_handedness
    uint32_t material;
    int32_t first_light;   // -1 none
    uint32_t alpha_tex1, shadow_alpha_tex1;  // 0, or 1 + the float texture that replaces the constant alpha / shadowalpha (triangle.rs:587-607, 868-898)
};
#define PH_MESH_N 1u
#define PH_MESH_S 2u
#define PH_MESH_UV 4u
#define PH_MESH_REV 8u
#define PH_MESH_SWAP 16u

// One BxDF of a material's BSDF (core/src/reflection/*.rs).  Constant textures make the list a property of the material, so the
// host builds it (api.hip) and the device walks it (bsdf_general.h).
enum { PH_LK_LAMBERT = 0, PH_LK_OREN = 1, PH_LK_SPEC_R = 2, PH_LK_SPEC_T = 3, PH_LK_FRESNEL_SPEC = 4, PH_LK_MICRO_R = 5, PH_LK_MICRO_T = 6,
       PH_LK_FRESNEL_BLEND = 7 /* r = Rd, t = Rs */, PH_LK_LAMBERT_T = 8 };
enum { PH_FR_NOOP = 0, PH_FR_DIEL = 1, PH_FR_COND = 2 };
#define PH_PRE_RAW_TEST 2u
struct alignas(16) LobeRec {
    uint32_t kind, type, fresnel, n_scale; // type = BxDFType bits (bsdf.rs:10-20); n_scale = ScaledBxDF wrappers around the lobe (mix.rs)
    float a, b;                           // Oren-Nayar A, B
    float ax, ay;                         // Trowbridge-Reitz alpha_x, alpha_y (already max(0.001, .))
    float eta_a, eta_b;                   // dielectric Fresnel (eta_i, eta_t) / transmission lobes (etaA, etaB)
    uint32_t eta_tex1, k_tex1;            // MetalMaterial: 0, or 1 + the texture behind the conductor's eta / k (metal.rs:121-125; evaluated per hit, NOT clamped)
    float r[3]; uint32_t r_tex1;          // r_tex1 / t_tex1: 0, or 1 + the texture that supplies this colour at every hit (set_material_texture)
    float t[3]; uint32_t t_tex1;
    float c_eta_t[3]; uint32_t amt;       // conductor Fresnel: eta_t and k (eta_i is ONE, metal.rs:84-88).  amt: MixMaterial with an `amount` texture: bits 0-1 = 1: this lobe's
    float c_k[3]; uint32_t alt;           //   scale[(amt >> 8) & 3] is the hit's s1, = 2: it is s2 = clamp(1 - s1) (mix.rs:59-60).  alt: GlassMaterial with roughness textures
    float scale0[3], ur_raw;              //   (glass.rs:110-141): 1 = the lobe of hits where urough == vrough == 0, 2 = a lobe of the other hits; ur_raw / vr_raw: the constant
    float scale1[3], vr_raw;              //   roughnesses before remapping.  scale0: innermost ScaledBxDF scale
    uint32_t ax_tex1, ay_tex1, remap, sigma_tex1;  // float textures for the Trowbridge-Reitz roughness (remapped per hit if `remap`) / MatteMaterial's sigma: 0 or 1 + id
    uint32_t slot0, pad0_[3];             // the first colour slot of TexOut this lobe's textured colours come from (set at upload: the slots taken by the lobes in front of it)
    float pre[3]; uint32_t has_pre;       // 1: a textured colour of this lobe is multiplied by `pre`, then tested (UberMaterial: `op * kd.evaluate().clamp_default()`, uber.rs:133);
                                          // 2: the texel is tested, then multiplied (TranslucentMaterial: `if !kd.is_black() { add(r * kd) }`, translucent.rs:77-84, :87) -- PH_PRE_RAW_TEST
                                          // 3: UberMaterial with an OPACITY TEXTURE: the colour is op(hit) * (texel, or the constant kept in `pre`) -- PH_PRE_OPACITY; 4: the pass-through
                                          //    lobe's colour clamp(1 - op(hit)) (uber.rs:126-160) -- PH_PRE_PASSTHROUGH.  Lobes of kind 3 / 4 always take a colour slot of the texture pass
};
#define PH_PRE_OPACITY 3u
#define PH_PRE_PASSTHROUGH 4u
#define PH_PRE_RT 5u   // TranslucentMaterial with a reflect / transmit texture (translucent.rs:70-98): colour = reflect-or-transmit(hit) * (texel, or the constant Kd / Ks kept in `pre`); both factors tested for black
#define PH_TEXOUT_NULL_BSDF (1u << 16)   // TexOut::bumped: this hit has no BSDF (reflect and transmit both black, translucent.rs:72-74): the path passes through like Material "none"

struct MaterialRec {  // matte fast path (materials/src/matte.rs with constant textures) + the general lobe list
    float kd[3];      // already clamp_default()'ed
    float sigma;      // already clamped to [0,90]; 0 -> LambertianReflection
    float a, b;       // OrenNayar A, B (core/src/reflection/oren_nayar.rs:28-39)
    uint32_t has_bxdf; // !kd.is_black()
    uint32_t lobe_base, n_lobes;  // DeviceScene::lobes[lobe_base .. lobe_base + n_lobes)
    uint32_t sort_class;  // 0..6: index of this material's lobe-kind signature among the scene's materials (block-local sorting key)
    float bsdf_eta;    // BSDF::eta (bsdf.rs:101): 1 unless the material passes one (uber.rs:131-138)
    uint32_t none;     // Material "none": no BSDF, the path integrator skips the surface (path.rs:142-150)
    uint32_t kd_tex1;  // 0, or 1 + the texture MatteMaterial evaluates for Kd at every hit (matte.rs:63): kd / has_bxdf are then per hit
    uint32_t textured; // some lobe of the list takes a colour from a texture: the general-BSDF kernel builds the hit's own list
    uint32_t bump_tex1; // 0, or 1 + the displacement texture of Material::bump (core/src/material.rs:62-101)
    uint32_t sigma_tex1; // MatteMaterial: 0, or 1 + the float texture behind sigma (the one-lobe kernel reads it here, the general one in its lobe)
    uint32_t opacity_tex1; // UberMaterial (or a mix holding one): 0, or 1 + the opacity texture; lobes with has_pre 3 / 4 depend on it
    uint32_t amount_tex1;  // MixMaterial: 0, or 1 + the `amount` texture (takes the FIRST colour slot of the texture pass)
    float bsdf_eta_alt;    // UberMaterial with an opacity texture: BSDF::eta of the hits that do not get the pass-through lobe (uber.rs:128-137)
    uint32_t uber_eta;     // 1: this material IS that uber (a mix holding one keeps eta 1)
    uint32_t rt_mode;      // TranslucentMaterial in the per-hit form (PH_PRE_RT lobes): reflect / transmit below are evaluated at every hit, a hit where both are black has no BSDF
    uint32_t refl_tex1, trans_tex1;   // 0, or 1 + the texture behind `reflect` / `transmit`
    float refl_c[3], trans_c[3];      // ... their clamped constants where they are not textures
    uint32_t index_tex1;   // GlassMaterial / UberMaterial: 0, or 1 + the float texture behind `index` (glass.rs:102, uber.rs:128): the dielectric lobes' eta of a hit (TexOut::col[2][3])
    uint32_t tex_cols;     // colour slots of TexOut the texture pass fills for this material (set at upload)
    uint32_t tex_hdr;      // the shade pass reads TexOut's header (bumped frame, per-hit scalars, lambert / glass / raw-black bits) for this material; else only the colours are written
};
// What the texture pass hands the shade pass for one path vertex (wavefront.hip: texture_kernel): the bumped shading frame and the evaluated,
// clamped (and pre-multiplied) colours of the material's textured lobe colours, in template-lobe order (r before t).
#define PH_HIT_COLS 6
struct TexOut { float ns[3]; uint32_t bumped; float dpdu_s[3]; uint32_t lambert; float col[PH_HIT_COLS][4]; };  // 128 B; col[0][3], col[1][3]: the per-hit (alpha_x, alpha_y) or Oren-Nayar (A, B); col[2][3]: the per-hit index of refraction; lambert: sigma evaluated to 0; bumped bit 8 + k: colour k's texel was black before its PH_PRE_RAW_TEST product
#define PH_HIT_LOBES 8   // lobes a material's template list may hold (= MAX_BXDFS, bsdf.rs:22: uber has up to 5, a mix up to 8): a hit's own list is a bit mask over them

// ---- textures (textures/src/*.rs, core/src/mipmap/mod.rs).  A texture is a postfix program over a small value stack: the host
// flattens the scale / mix tree once (api.hip), the device runs it per hit (texture.h).  Float-valued textures are carried as three equal
// channels; the one place the reference treats them differently (the final division of MIPMap::ewa) is selected by MipRec::is_float.
enum { PH_TOP_CONST = 0, PH_TOP_IMAGE = 1, PH_TOP_MUL = 2, PH_TOP_MIX = 3, PH_TOP_CHECKER = 4 /* pops tex1, tex2; mip = aa mode */, PH_TOP_UV = 5,
       PH_TOP_BILERP = 6 /* pops v00 v01 v10 v11 */, PH_TOP_DOTS = 7 /* pops inside, outside */, PH_TOP_FBM = 8, PH_TOP_WRINKLED = 9, PH_TOP_WINDY = 10, PH_TOP_MARBLE = 11,
       PH_TOP_CHECKER3D = 12 /* pops tex1, tex2 */ };
#define PH_TEX_STACK 6
struct TexOp {
    uint32_t op;
    uint32_t mip;          // PH_TOP_IMAGE: index into DeviceScene::mipmaps
    float c[3];            // PH_TOP_CONST
    float su, sv, du, dv;  // 2D textures: UVMapping2D (texture/mapping/uv_2d.rs)
    uint32_t octaves;      // fbm / wrinkled / marble
    float omega, scale, variation;
    float m[16];           // 3D textures: IdentityMapping3D's matrix (texture/mapping/identity_3d.rs); 2D textures with mapping 1 / 2: world_to_texture; 3: vs, vt
    uint32_t mapping;      // 2D textures: 0 uv, 1 spherical, 2 cylindrical, 3 planar (du, dv = ds, dt)
    uint32_t pad_[3];
};
struct TexRec { uint32_t first_op, n_ops; };  // DeviceScene::tex_ops[first_op .. first_op + n_ops)
struct Texel { float r, g, b, pad; };  // one 16-byte load per texel
#define PH_MIP_MAX_LEVELS 16
struct MipRec {  // MIPMap<T> (core/src/mipmap/mod.rs:76-96): pyramid levels in one pool of float4 texels (rgb + pad), row-major [t][s]
    uint32_t n_levels, filtering /*0 trilinear 1 EWA*/, wrap /*0 repeat 1 black 2 clamp*/, is_float;
    float max_anisotropy;
    uint32_t pad[3];
    uint32_t level_off[PH_MIP_MAX_LEVELS];  // first texel of each level in DeviceScene::texels
    uint32_t level_w[PH_MIP_MAX_LEVELS], level_h[PH_MIP_MAX_LEVELS];
};
#define PH_EWA_LUT_SIZE 128

enum { PH_L_INFINITE = 0, PH_L_DISTANT = 1, PH_L_POINT = 2, PH_L_AREA = 3, PH_L_SPOT = 4, PH_L_PROJECTION = 5, PH_L_GONIO = 6 };
struct LightRec {
    int32_t type;
    int32_t two_sided;
    uint32_t prim;       // area: bound triangle
    float area;          // area: Triangle::area()
    float L[3];          // radiance / intensity
    uint32_t map_mip1;   // infinite: 0, or 1 + the MIPMap of its radiance map (`mapname`); then the Distribution2D lives in DeviceScene::light_dist
    float v[3];          // distant: w_light; point / spot: p_light
    uint32_t dist_off;   // first float of {cond_func[dh][dw], cond_cdf[dh][dw+1], cond_int[dh], marg_func[dh], marg_cdf[dh+1], marg_int}
    float l2w[12];       // infinite: rows 0..2 of light_to_world (3x4)
    float w2l[12];       // infinite: rows 0..2 of world_to_light
    // infinite: Distribution2D over the 2x2 scalar image (lights/src/infinite.rs:326-369)
    float cond_func[4], cond_cdf[6], cond_int[2];
    float marg_func[2], marg_cdf[3], marg_int;
    float cos_total_width, cos_falloff_start;  // spot (lights/src/spot.rs:24-25); w2l holds its world_to_light, v its position
    uint32_t dw, dh;     // radiance map: resolution of the scalar image = 2 x the map's
    float proj[16];      // projection light: light_projection (lights/src/projection.rs:104); screen = its screen_bounds {xmin, xmax, ymin, ymax};
    float screen[4];     // map_mip1 = the projected image / the goniometric diagram (0: none), w2l = world_to_light, v = p_light
};

enum { PH_CAM_PERSPECTIVE = 0, PH_CAM_ORTHOGRAPHIC = 1, PH_CAM_ENVIRONMENT = 2 };
struct CameraRec {  // cameras/src/perspective_camera.rs, orthographic_camera.rs, environment_camera.rs
    float r2c[16];
    float c2w[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
    float dx_camera[3], dy_camera[3];  // :70-74, for the ray differentials texture filtering needs
    uint32_t kind; float pad[1];
    float full_res[2];  // EnvironmentCamera: film.full_resolution as floats (environment_camera.rs:63-64)
};

struct FilmRec {  // core/src/film/mod.rs
    int32_t xres, yres;
    int32_t crop[4];
    float radius[2], inv_radius[2];
    float scale, max_lum;
    float table[256];
};

struct SamplerRec {  // samplers/src/halton.rs:61-100
    int32_t kind;  // 0 halton, 1 sobol
    uint32_t spp;
    int32_t bounds[4];
    int32_t at_center;
    // halton
    uint32_t base_scales[2], base_exponents[2], sample_stride;
    uint32_t mult_inverse[2];
    // sobol
    int32_t resolution, log2_resolution;
};

// Everything a kernel needs, passed by value (fits the 4 KB kernarg budget comfortably).
struct DeviceScene {
    const Node64* nodes;
    const TriRec* tris;
    uint32_t root_ref;        // PH_INVALID_REF when the scene is empty
    uint32_t n_tris;
    float root_lo[3], root_hi[3];
    float world_center[3], world_radius;
    // shading geometry
    const float* P;           // 3 floats per vertex
    const float* N;           // may be null
    const float* S;
    const float* UV;
    const uint32_t* idx;      // 3 per triangle (global vertex ids)
    const uint32_t* tri_mesh; // mesh id per triangle
    const uint32_t* tri_flags; // PH_TRI_BOGUS/ALPHA0/SALPHA0 per triangle (add_mesh order)
    const MeshRec* meshes;
    const MaterialRec* materials;
    const LobeRec* lobes;
    const TexRec* textures;
    const TexOp* tex_ops;
    const MipRec* mipmaps;
    const Texel* texels;
    const float* ewa_lut;     // PH_EWA_LUT_SIZE Gaussian weights (mipmap/mod.rs:168-174), made on the host with its expf
    const float* light_dist;  // Distribution2D tables of infinite lights with a radiance map
    const DeviceScene* self;  // device-resident copy of this struct: out-of-line device functions take it instead of the by-value kernel argument
    const LightRec* lights;
    uint32_t n_lights;
    const uint32_t* infinite_lights;
    uint32_t n_infinite;
    const uint16_t* mat_key;    // per material: its key in the shade-side work queues (matsort.h): materials with per-hit textures / a bump map hold keys [0, ms_tex_keys), the others follow,
    uint32_t ms_tex_keys, ms_key_emit, ms_key_idle;   // then ms_key_emit (new vertex at the bounce limit: emission only) and ms_key_idle (no new vertex)
    const InstRec* instances;
    const float* inst_extra;    // 12 floats (read as 3 float4) per scene-level leaf record: rows 0 .. 2 of world-to-instance where the record is an instance (else unused)
    uint32_t n_instances;
    // light-selection Distribution1D (core/src/sampling/distribution_1d.rs)
    const float* ld_func;
    const float* ld_cdf;      // n_lights + 1
    float ld_func_int;
    // sampler tables
    const uint16_t* halton_perms;
    const uint32_t* primes;       // 1000
    const uint32_t* prime_sums;   // 1000
    const uint64_t* prime_magic;  // 1000: ceil(2^64/p), exact u32 division by multiply-high
    const uint32_t* sobol32;
    const uint64_t* vdc;
    const uint64_t* vdc_inv;
};
