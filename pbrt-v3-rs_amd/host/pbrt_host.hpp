// C++ host side above the C ABI: the reference's scene-description API (api/src/lib.rs) for the subset of directives the
// hot path can consume, feeding include/pbrt_hip.h.  Method names, argument meaning, option/world block rules and CTM
// semantics follow the reference's `Api` (api/src/lib.rs:85-1075) so this file reads like the Rust host it stands in for
// (the image has no Rust toolchain).  Nothing here computes radiance: it only captures the scene and calls the library.
#pragma once
#include "../../include/pbrt_hip.h"
#include "../../include/pbrt_hip_host.h"
#include <array>
#include <map>
#include <string>
#include <vector>

namespace pbrt_host {

// core/src/paramset/mod.rs: typed parameter bags with defaults at the point of use
struct ParamSet {
    std::map<std::string, std::vector<float>> floats;      // float, point*, vector*, normal*, rgb/color
    std::map<std::string, std::vector<int>> ints;
    std::map<std::string, std::vector<std::string>> strings;
    std::map<std::string, std::vector<std::string>> textures;  // "texture name" references
    std::vector<std::string> unsupported;                      // parameters of a type this host cannot evaluate (spectrum/blackbody)
    std::map<std::string, std::vector<bool>> bools;
    float find_one_float(const std::string& n, float d) const;
    int find_one_int(const std::string& n, int d) const;
    bool find_one_bool(const std::string& n, bool d) const;
    std::string find_one_string(const std::string& n, const std::string& d) const;
    std::string find_one_texture(const std::string& n) const;  // "" when absent
    const std::vector<float>* find_floats(const std::string& n) const;
    const std::vector<int>* find_ints(const std::string& n) const;
    std::array<float, 3> find_one_rgb(const std::string& n, std::array<float, 3> d) const;
};

struct Xform { float m[16], mi[16]; };  // Transform {m, m_inv} (core/src/geometry/transform.rs:12-18)

struct MaterialDesc { std::string type = "matte"; ParamSet params; };

struct GraphicsState {  // api/src/graphics_state.rs:60-130
    MaterialDesc material;
    std::string area_light;  // "" = none
    ParamSet area_light_params;
    bool reverse_orientation = false;
    std::map<std::string, MaterialDesc> named_materials;
    std::map<std::string, std::array<float, 3>> spectrum_textures;  // constant textures only
    std::map<std::string, float> float_textures;
    std::map<std::string, std::string> unsupported_textures;       // name -> class, reported only if something uses them
    struct DeviceTexture { bool is_float; uint32_t id; };          // imagemap, or scale / mix over one: evaluated by the library per hit
    std::map<std::string, DeviceTexture> device_textures;
};

struct RenderReport {
    PbrtHipStats stats{};
    int xres = 0, yres = 0, crop[4] = {0, 0, 0, 0};
    std::string out_file;
    double build_seconds = 0, load_seconds = 0;
    uint64_t n_triangles = 0, n_lights = 0, n_instances = 0;  // n_triangles counts instanced triangles once per instance
    int spp = 0, max_depth = 0, light_strategy = 0, pixel_bounds[4] = {0, 0, 0, 0};
    std::vector<std::string> warnings;
};

class Api {
  public:
    explicit Api(int device = 0);
    explicit Api(const std::vector<int>& devices);   // one handle over several GPUs (pbrt_hip_scene_create_multi); empty = every visible device
    ~Api();
    Api(const Api&) = delete;
    Api& operator=(const Api&) = delete;
    // --- transformations (api/src/lib.rs:132-326)
    void pbrt_identity();
    void pbrt_translate(float dx, float dy, float dz);
    void pbrt_rotate(float angle, float dx, float dy, float dz);
    void pbrt_scale(float sx, float sy, float sz);
    void pbrt_look_at(float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy, float uz);
    void pbrt_concat_transform(const float tr[16]);
    void pbrt_transform(const float tr[16]);
    void pbrt_coordinate_system(const std::string& name);
    void pbrt_coord_sys_transform(const std::string& name);
    // --- options block (:328-392)
    void pbrt_pixel_filter(const std::string& name, const ParamSet& p);
    void pbrt_film(const std::string& type, const ParamSet& p);
    void pbrt_sampler(const std::string& name, const ParamSet& p);
    void pbrt_accelerator(const std::string& name, const ParamSet& p);
    void pbrt_integrator(const std::string& name, const ParamSet& p);
    void pbrt_camera(const std::string& name, const ParamSet& p);
    // --- world block (:434-1000)
    void pbrt_world_begin();
    void pbrt_attribute_begin();
    void pbrt_attribute_end();
    void pbrt_transform_begin();
    void pbrt_transform_end();
    void pbrt_texture(const std::string& name, const std::string& type, const std::string& tex_class, const ParamSet& p, const std::string& scene_dir = "");
    void pbrt_material(const std::string& name, const ParamSet& p);
    void pbrt_make_named_material(const std::string& name, const ParamSet& p);
    void pbrt_named_material(const std::string& name);
    void pbrt_light_source(const std::string& name, const ParamSet& p, const std::string& scene_dir = "");
    void pbrt_area_light_source(const std::string& name, const ParamSet& p);
    void pbrt_shape(const std::string& name, const ParamSet& p, const std::string& scene_dir);
    void pbrt_reverse_orientation();
    void pbrt_object_begin(const std::string& name);
    void pbrt_object_end();
    void pbrt_object_instance(const std::string& name);
    // builds the accelerator, renders, writes the image; returns 0 or a PBRT_HIP_ERR_* code
    int pbrt_world_end(RenderReport& report);

    // command-line overrides (core/src/app/options.rs:13-88)
    std::string override_outfile;
    bool has_crop_override = false;
    float crop_override[4] = {0, 1, 0, 1};  // --cropwindow x0 x1 y0 y1
    std::string sobol_tables_file;  // raw little-endian: u32[1024*52], u64[25*52], u64[26*52]
    int tile_size = 16;
    bool quiet = false;
    std::vector<std::string> warnings;
    std::string error;

    void warn(const std::string& w);   // also used by the parser (unreadable SPD files leave a black spectrum with a warning, paramset/mod.rs:296-299)
  private:
    bool check_only_ = false;
    bool world_block_ = false;
    PbrtHipScene* scene_ = nullptr;
    Xform ctm_;
    std::map<std::string, Xform> named_cs_;
    std::vector<GraphicsState> gs_stack_;
    std::vector<Xform> ctm_stack_;
    GraphicsState gs_;
    // stashed options (constructed lazily at WorldEnd like the reference, api/src/lib.rs:447-507)
    std::string filter_name_ = "box", film_name_ = "image", sampler_name_ = "halton", accel_name_ = "bvh", integrator_name_ = "path", camera_name_ = "perspective";
    ParamSet filter_p_, film_p_, sampler_p_, accel_p_, integrator_p_, camera_p_;
    Xform camera_to_world_;
    uint64_t n_tris_ = 0, n_lights_ = 0;
    std::map<std::string, uint32_t> material_cache_;
    std::map<std::string, uint32_t> mipmap_cache_;  // MIPMapCache (core/src/mipmap/cache.rs:28-60), keyed by TexInfo + texel type
    std::map<std::string, uint32_t> objects_;   // named object instances (render_options.instances)
    std::map<std::string, uint64_t> object_tris_;
    std::string current_object_;                // "" outside ObjectBegin/ObjectEnd
    uint64_t n_instances_ = 0;
    bool verify_options(const char* func);
    bool verify_world(const char* func);
    void concat(const Xform& t);
    uint32_t material_id_for(const MaterialDesc& m);
    bool check(int rc, const char* what);
};

// api/src/parser: the subset of the .pbrt grammar listed in SURVEY Appendix E.  Returns false and sets api.error on failure.
bool parse_file(const std::string& path, Api& api, RenderReport* report_out);
bool parse_string(const std::string& text, const std::string& scene_dir, Api& api, RenderReport* report_out);

// shapes/src/plymesh.rs:165-249 — vertices (x y z [nx ny nz] [u v | s t]) and faces of 3 or 4 vertex_indices
struct PlyMesh { std::vector<float> P, N, UV; std::vector<uint32_t> indices; };
bool read_ply(const std::string& path, PlyMesh& out, std::string& err);

// core/src/image_io.rs:42-50 (PFM, TGA, PNG): width*height RGB floats, top row first
bool read_image(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err);

// core/src/image_io.rs:225-237: .pfm, .exr (uncompressed float), .png / .tga (8-bit through apply_gamma)
bool write_image(const std::string& path, const float* rgb, int w, int h, std::string& err);

// core/src/image_io.rs:336-374
bool write_pfm(const std::string& path, const float* rgb, int w, int h, std::string& err);

}  // namespace pbrt_host
