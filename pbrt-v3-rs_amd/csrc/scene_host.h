// Host-side scene object behind the opaque PbrtHipScene handle (include/pbrt_hip.h).
#pragma once
#include "../../include/pbrt_hip.h"
#include "bvh_build.h"
#include "guard.h"
#include "scene_types.h"
#include <hip/hip_runtime.h>
namespace ph { struct TravParams; }
#include <string>
#include <vector>

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Everything the capture calls record and build_accel derives, i.e. the part of a scene that does not belong to a device.  A multi-device handle
// (pbrt_hip_scene_create_multi) copies it as a whole to its per-device contexts (multi.hip).
struct SceneHostState {
    // ---- captured scene (host copies, add_mesh order) ----------------------------------------------------------------
    std::vector<float> P, N, S, UV;  // N/S/UV are vertex-aligned with P when any mesh has them (zero-filled otherwise)
    bool any_n = false, any_s = false, any_uv = false;
    std::vector<uint32_t> idx, tri_mesh, tri_flags;
    std::vector<MeshRec> meshes;
    std::vector<MaterialRec> materials;
    std::vector<LobeRec> lobes;
    // textures: every texture id owns a flattened postfix program (its children's programs followed by its own op)
    struct TextureHost { std::vector<TexOp> prog; int stack_need = 1; };
    std::vector<TextureHost> textures;
    std::vector<MipRec> mipmaps;
    std::vector<Texel> texels;
    // which lobe and which of its two colours each texturable parameter of a material feeds (set_material_texture); -1 = that lobe was not made
    struct MaterialParams { int lobe[4] = {-1, -1, -1, -1}; int field[4] = {0, 0, 0, 0}; int lobe2[4] = {-1, -1, -1, -1}; int field2[4] = {0, 0, 0, 0}; bool has_pre = false; float pre[3] = {1, 1, 1}; int rough_lobe = -1, rough_lobe2 = -1; bool rough_remap = false;
                            // what the material was made from, for the setters that rebuild the lobe list (uber opacity / glass roughness textures), and the mix's split
                            int made_as = 0;  /* 0 other, 1 uber, 2 glass, 3 metal, 4 mix, 5 translucent */ float raw_k[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; float raw_eta = 1.5f, raw_ur = 0, raw_vr = 0;
                            bool raw_remap = false, rebuilt = false; int mix_n1 = 0; };  // [Kd, Ks, Kr, Kt]; field 0 = r, 1 = t; pre: uber's opacity; lobe2 / field2 / rough_lobe2: a second lobe fed by the same parameter (translucent's reflection + transmission pair); rough_lobe: owner of the Trowbridge-Reitz distribution
    std::vector<MaterialParams> material_params;
    bool alpha_textures = false;      // some mesh has an alpha / shadowalpha texture: traversal uses the ALPHA kernel variants
    bool alpha_lean = false;          // ... and all of them are constants / uv-mapped image maps / scale / mix: the inlined test (set at upload)
    bool bump_materials = false;      // some material has a bump map
    bool textured_materials = false;  // some material evaluates a texture per hit
    bool has_none_material = false;  // some material is "none": paths may need more wavefront iterations than max_depth + 1
    bool general_materials = false;  // some material is not matte: the renderer uses the general BSDF kernel
    bool simple_textures = false;    // every texture program is made of constants, image maps (uv mapping), scale and mix: the texture pass runs its lean instantiation (set at upload)
    std::vector<LightRec> lights;
    // DiffuseAreaLights of shapes inside an object definition ("Area lights not supported with object instancing", api/src/lib.rs:877-881): the primitives keep their emission,
    // Scene::lights never sees them.  MeshRec::first_light = -2 - index on the host; uploaded behind the scene's lights (DeviceScene::n_lights does not count them).
    std::vector<LightRec> emission_only;
    std::vector<uint32_t> infinite_lights;
    std::vector<float> light_dist;   // Distribution2D tables of infinite lights with a radiance map
    // object instancing (api/src/lib.rs:911-1000): an object is a contiguous triangle range; top_items is the scene's primitive
    // list in directive order (triangle id, or PH_ITEM_INST | instance index)
    struct ObjectHost { uint32_t tri0 = 0, tri1 = 0; };
    struct InstanceHost { uint32_t object; float i2w[16], w2i[16]; };
    std::vector<ObjectHost> objects;
    std::vector<InstanceHost> instances;
    std::vector<uint32_t> top_items;
    int open_object = -1;
    std::vector<InstRec> inst_recs;  // built by build_accel
    CameraRec cam{};
    FilmRec film{};
    SamplerRec sampler{};
    bool have_camera = false, have_film = false, have_sampler = false, built = false;
    std::vector<uint32_t> sobol32;
    std::vector<uint64_t> vdc, vdc_inv;

    // ---- acceleration structure ---------------------------------------------------------------------------------------
    phost::BuildOutput bvh;
    float world_center[3] = {0, 0, 0}, world_radius = 1.0f;
};

struct MultiDevice;  // multi.hip

struct PbrtHipScene : SceneHostState {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool build_on_device = false;    // set for the duration of pbrt_hip_build_accel_device
    MultiDevice* multi = nullptr;    // non-null on a handle made by pbrt_hip_scene_create_multi: the other devices' contexts and the exchange state

    // ---- device residency ---------------------------------------------------------------------------------------------
    DeviceScene ds{};
    std::vector<void*> owned;        // every hipMalloc of the scene, freed on destroy / rebuild
    // a tree pbrt_hip_build_accel_device(0, ..) left where it was built: `bvh.nodes` / `bvh.tris` stay empty on the host (accel_copy reads them back on request; the multi-device driver replicates by hipMemcpyPeer)
    void* tree_dev_nodes = nullptr; void* tree_dev_tris = nullptr; size_t tree_dev_n_tris = 0;
    bool uploaded = false;
    int light_strategy_uploaded = -1;
    DevBuf d_ld_func, d_ld_cdf;

    // traversal workspace
    DevBuf d_counter, d_spill, d_error, d_rays_tmp, d_out_tmp;
    uint32_t trav_blocks = 0;
    bool count_traversal = false;    // roofline bookkeeping mode (pbrt_hip_set_traversal_counting)
    DevBuf d_counts;                 // 8 u64: closest-hit {nodes, tris, rays}, any-hit {nodes, tris, rays}, any-hit reference node visits, spare
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // wavefront workspace (allocated lazily by the renderer, see wavefront.hip)
    struct Wavefront* wf = nullptr;
};

namespace phost {
int set_err(PbrtHipScene* s, int code, const std::string& msg);
int hip_fail(PbrtHipScene* s, hipError_t e, const char* what);
int ensure_buf(PbrtHipScene* s, DevBuf& b, size_t bytes);
int upload_scene(PbrtHipScene* s);
void projection_light_setup(float fov_deg, float aspect, float out_proj[16], float out_screen[4], float* out_cos_total_width);   // host_setup.cpp
// MIPMap::lookup_triangle on the host copy of the texel pool (textures_api.hip), for what InfiniteAreaLight computes at construction time
void hmip_lookup_triangle(const PbrtHipScene* s, const MipRec& m, float u, float v, float width, float out[3]);
int upload_light_distribution(PbrtHipScene* s, int light_strategy);
int launch_traverse(PbrtHipScene* s, bool anyhit, const void* d_rays, void* d_out, uint32_t n, float* kernel_ms);
void launch_traverse_kernel(PbrtHipScene* s, int mode, uint32_t blocks, const ph::TravParams& p);  // 0 closest, 1 any hit, 2 both (MIXED)
int ensure_traversal_workspace(PbrtHipScene* s);
void free_wavefront(PbrtHipScene* s);
int build_hlbvh_device(const BuildInput& in, int max_prims_in_node, hipStream_t stream, BuildOutput& out, std::string& err);   // bvh_device.hip
// bvh_sah_device.hip.  keep_nodes / keep_tris non-null: the Node64 / TriRec arrays are not copied to `out` but handed over as device allocations (the caller frees them)
// forest non-null: the scene's aggregate and its instanced objects' aggregates in one build, laid out as build_forest_host lays them out; trees_out: every tree's root
int build_sah_device(const BuildInput& in, int max_prims_in_node, hipStream_t stream, BuildOutput& out, std::string& err, void** keep_nodes = nullptr, void** keep_tris = nullptr,
                     const ForestSpec* forest = nullptr, std::vector<ForestTreeOut>* trees_out = nullptr);
void free_tree_dev(PbrtHipScene* s);
int uber_rebuild_for_opacity(PbrtHipScene* s, uint32_t material);    // api.hip: lobe lists remade when a structural parameter becomes a texture
int glass_rebuild_for_roughness(PbrtHipScene* s, uint32_t material);
int translucent_rebuild_rt(PbrtHipScene* s, uint32_t material);      // api.hip: TranslucentMaterial -> its per-hit form (a reflect / transmit texture)
// wavefront.hip: the renderer's building blocks, shared with the multi-device driver (multi.hip)
#define PH_MAX_TILE_PARTS 64
int check_render_args(PbrtHipScene* s, int max_depth, int light_strategy, const int* pixel_bounds, int tile_size, int part, int parts);
size_t tile_buffer_floats_for(const PbrtHipScene* s, int tile_size, int part, int parts);
int render_tiles(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size, int part, int parts, void* d_tile_buffer,
                 PbrtHipStats* out_stats);
int merge_tiles(PbrtHipScene* s, int tile_size, int parts, const void* const* d_bufs, float* out_xyz, float* out_weight);
DevBuf& tile_buffer_of(PbrtHipScene* s);  // the handle's own tile buffer (wavefront workspace)
// ORs the voxels whose light distribution the context's last spatial render filled into `touched` (one byte per voxel, sized on first use); *count = voxels set so far
int spatial_voxels_touched(PbrtHipScene* s, std::vector<uint8_t>& touched, uint64_t* count);
// multi.hip
int render_path_multi(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size, int tile_part, int tile_parts,
                      float* out_xyz, float* out_weight, PbrtHipStats* out_stats);
void free_multi(PbrtHipScene* s);
}  // namespace phost

#define PH_CHECK(s, call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return phost::hip_fail((s), e__, #call); \
    } while (0)
