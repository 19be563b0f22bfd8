// Host-side scene API over the C ABI.  See pbrt_host.hpp for scope; reference: api/src/lib.rs, api/src/graphics_state.rs.
#include "pbrt_host.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>

namespace pbrt_host {

// ------------------------------------------------------------------------------------------------ ParamSet
float ParamSet::find_one_float(const std::string& n, float d) const {
    auto it = floats.find(n);
    return (it != floats.end() && it->second.size() == 1) ? it->second[0] : d;  // paramset: one-value lookups need exactly one
}
int ParamSet::find_one_int(const std::string& n, int d) const {
    auto it = ints.find(n);
    return (it != ints.end() && it->second.size() == 1) ? it->second[0] : d;
}
bool ParamSet::find_one_bool(const std::string& n, bool d) const {
    auto it = bools.find(n);
    return (it != bools.end() && it->second.size() == 1) ? it->second[0] : d;
}
std::string ParamSet::find_one_string(const std::string& n, const std::string& d) const {
    auto it = strings.find(n);
    return (it != strings.end() && it->second.size() == 1) ? it->second[0] : d;
}
std::string ParamSet::find_one_texture(const std::string& n) const {
    auto it = textures.find(n);
    return (it != textures.end() && it->second.size() == 1) ? it->second[0] : std::string();
}
const std::vector<float>* ParamSet::find_floats(const std::string& n) const {
    auto it = floats.find(n);
    return it == floats.end() ? nullptr : &it->second;
}
const std::vector<int>* ParamSet::find_ints(const std::string& n) const {
    auto it = ints.find(n);
    return it == ints.end() ? nullptr : &it->second;
}
std::array<float, 3> ParamSet::find_one_rgb(const std::string& n, std::array<float, 3> d) const {
    auto it = floats.find(n);
    if (it == floats.end() || it->second.size() != 3) return d;
    return {it->second[0], it->second[1], it->second[2]};
}

// ------------------------------------------------------------------------------------------------ Api
static Xform identity_xform() {
    Xform x{};
    for (int i = 0; i < 4; i++) x.m[i * 5] = x.mi[i * 5] = 1.0f;
    return x;
}

// device < 0 = check mode: directives are parsed, validated and counted, nothing is sent to the library and nothing is
// rendered (there is no CPU renderer to fall back to).
#define ABI(call) (check_only_ ? (int)PBRT_HIP_OK : (call))

Api::Api(int device) : check_only_(device < 0), ctm_(identity_xform()), camera_to_world_(identity_xform()) {
    if (check_only_) return;
    scene_ = pbrt_hip_scene_create(device);
    if (!scene_) {
        const char* e = pbrt_hip_last_error(nullptr);
        error = std::string("no usable gfx950 device: ") + (e ? e : "");
    }
}
Api::Api(const std::vector<int>& devices) : check_only_(false), ctm_(identity_xform()), camera_to_world_(identity_xform()) {
    scene_ = pbrt_hip_scene_create_multi(devices.empty() ? nullptr : devices.data(), (int)devices.size());
    if (!scene_) {
        const char* e = pbrt_hip_last_error(nullptr);
        error = std::string("no usable gfx950 device: ") + (e ? e : "");
    }
}
Api::~Api() {
    if (scene_) pbrt_hip_scene_destroy(scene_);
}

void Api::warn(const std::string& w) {
    warnings.push_back(w);
    if (!quiet) std::fprintf(stderr, "Warning: %s\n", w.c_str());
}
bool Api::check(int rc, const char* what) {
    if (rc == PBRT_HIP_OK) return true;
    if (error.empty()) {
        const char* e = scene_ ? pbrt_hip_last_error(scene_) : nullptr;
        error = std::string(what) + " failed (" + std::to_string(rc) + "): " + (e ? e : "");
    }
    return false;
}
// `ctm = ctm * t` for every active transform (api/src/lib.rs:140-152); the time-1 copy is dead on this path.
void Api::concat(const Xform& t) {
    Xform r;
    pbrt_hip_host_compose(ctm_.m, ctm_.mi, t.m, t.mi, r.m, r.mi);
    ctm_ = r;
}

void Api::pbrt_identity() { ctm_ = identity_xform(); }
void Api::pbrt_translate(float dx, float dy, float dz) {
    Xform t; const float d[3] = {dx, dy, dz};
    pbrt_hip_host_translate(d, t.m, t.mi);
    concat(t);
}
void Api::pbrt_rotate(float angle, float dx, float dy, float dz) {
    Xform t; const float a[3] = {dx, dy, dz};
    pbrt_hip_host_rotate(angle, a, t.m, t.mi);
    concat(t);
}
void Api::pbrt_scale(float sx, float sy, float sz) {
    Xform t; const float s[3] = {sx, sy, sz};
    pbrt_hip_host_scale(s, t.m, t.mi);
    concat(t);
}
void Api::pbrt_look_at(float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy, float uz) {
    Xform t; const float e[3] = {ex, ey, ez}, l[3] = {lx, ly, lz}, u[3] = {ux, uy, uz};
    if (pbrt_hip_host_look_at(e, l, u, t.m, t.mi) != 0) {
        // transform.rs:173-181: "up" and the viewing direction are collinear -> identity with an error message
        warn("LookAt: up vector and viewing direction are pointing in the same direction; using the identity transformation");
        t = identity_xform();
    }
    concat(t);
}
// The file format hands matrices over column-major; Matrix4x4::new takes rows (api/src/lib.rs:208-262).
static Xform from_column_major(const float tr[16]) {
    Xform t;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t.m[r * 4 + c] = tr[c * 4 + r];
    pbrt_hip_host_invert(t.m, t.mi);
    return t;
}
void Api::pbrt_concat_transform(const float tr[16]) { concat(from_column_major(tr)); }
void Api::pbrt_transform(const float tr[16]) { ctm_ = from_column_major(tr); }
void Api::pbrt_coordinate_system(const std::string& name) { named_cs_[name] = ctm_; }
void Api::pbrt_coord_sys_transform(const std::string& name) {
    auto it = named_cs_.find(name);
    if (it != named_cs_.end()) ctm_ = it->second;
    else warn("Couldn't find named coordinate system '" + name + "'");
}

// verify_options / verify_world (api/src/lib.rs:1003-1050): a directive in the wrong block is reported and ignored.
bool Api::verify_options(const char* func) {
    if (!world_block_) return true;
    warn(std::string("Options cannot be set inside world block; '") + func + "' not allowed. Ignoring.");
    return false;
}
bool Api::verify_world(const char* func) {
    if (world_block_) return true;
    warn(std::string("Scene description must be inside world block; '") + func + "' not allowed. Ignoring.");
    return false;
}

void Api::pbrt_pixel_filter(const std::string& name, const ParamSet& p) { if (verify_options("PixelFilter")) { filter_name_ = name; filter_p_ = p; } }
void Api::pbrt_film(const std::string& type, const ParamSet& p) { if (verify_options("Film")) { film_name_ = type; film_p_ = p; } }
void Api::pbrt_sampler(const std::string& name, const ParamSet& p) { if (verify_options("Sampler")) { sampler_name_ = name; sampler_p_ = p; } }
void Api::pbrt_accelerator(const std::string& name, const ParamSet& p) { if (verify_options("Accelerator")) { accel_name_ = name; accel_p_ = p; } }
void Api::pbrt_integrator(const std::string& name, const ParamSet& p) { if (verify_options("Integrator")) { integrator_name_ = name; integrator_p_ = p; } }
void Api::pbrt_camera(const std::string& name, const ParamSet& p) {
    if (!verify_options("Camera")) return;
    camera_name_ = name; camera_p_ = p;
    // camera_to_world = inverse(ctm): swap m and m_inv (api/src/lib.rs:376-392)
    std::memcpy(camera_to_world_.m, ctm_.mi, sizeof ctm_.mi);
    std::memcpy(camera_to_world_.mi, ctm_.m, sizeof ctm_.m);
    named_cs_["camera"] = camera_to_world_;
}

void Api::pbrt_world_begin() {
    if (!verify_options("WorldBegin")) return;
    world_block_ = true;
    ctm_ = identity_xform();
    named_cs_["world"] = ctm_;
}
void Api::pbrt_attribute_begin() { if (!verify_world("AttributeBegin")) return; gs_stack_.push_back(gs_); ctm_stack_.push_back(ctm_); }
void Api::pbrt_attribute_end() {
    if (!verify_world("AttributeEnd")) return;
    if (gs_stack_.empty()) { warn("Unmatched AttributeEnd encountered. Ignoring it."); return; }
    gs_ = gs_stack_.back(); gs_stack_.pop_back();
    ctm_ = ctm_stack_.back(); ctm_stack_.pop_back();
}
void Api::pbrt_transform_begin() { if (!verify_world("TransformBegin")) return; ctm_stack_.push_back(ctm_); }
void Api::pbrt_transform_end() {
    if (!verify_world("TransformEnd")) return;
    if (ctm_stack_.size() <= gs_stack_.size()) { warn("Unmatched TransformEnd encountered. Ignoring it."); return; }
    ctm_ = ctm_stack_.back(); ctm_stack_.pop_back();
}

void Api::pbrt_texture(const std::string& name, const std::string& type, const std::string& tex_class, const ParamSet& p, const std::string& scene_dir) {
    if (!verify_world("Texture")) return;
    const bool is_float = type == "float", is_spec = type == "color" || type == "spectrum";
    if (!is_float && !is_spec) { warn("Texture type '" + type + "' unknown."); return; }
    // "scale" and "mix" of constant textures are constant: fold them (textures/src/scale.rs:38-40 tex1 * tex2; mix.rs:44-49 (1 - amt) * t1 + amt * t2)
    if (tex_class == "scale" || tex_class == "mix") {
        bool ok = true;
        auto fget = [&](const char* n, float d) {
            std::string tn = p.find_one_texture(n);
            if (!tn.empty()) { auto it = gs_.float_textures.find(tn); if (it != gs_.float_textures.end()) return it->second; ok = false; }
            return p.find_one_float(n, d);
        };
        auto sget = [&](const char* n, std::array<float, 3> d) {
            std::string tn = p.find_one_texture(n);
            if (!tn.empty()) { auto it = gs_.spectrum_textures.find(tn); if (it != gs_.spectrum_textures.end()) return it->second; ok = false; }
            return p.find_one_rgb(n, d);
        };
        if (is_float) {
            const float t1 = fget("tex1", tex_class == "scale" ? 1.0f : 0.0f), t2 = fget("tex2", 1.0f);
            const float v = tex_class == "scale" ? t1 * t2 : ((1.0f - fget("amount", 0.5f)) * t1 + fget("amount", 0.5f) * t2);
            if (ok) { gs_.unsupported_textures.erase(name); gs_.device_textures.erase(name); gs_.float_textures[name] = v; return; }
        } else {
            const std::array<float, 3> t1 = sget("tex1", tex_class == "scale" ? std::array<float, 3>{1, 1, 1} : std::array<float, 3>{0, 0, 0}), t2 = sget("tex2", {1, 1, 1});
            std::array<float, 3> v;
            const float amt = fget("amount", 0.5f);
            for (int c = 0; c < 3; c++) v[c] = tex_class == "scale" ? t1[c] * t2[c] : ((1.0f - amt) * t1[c] + amt * t2[c]);
            if (ok) { gs_.unsupported_textures.erase(name); gs_.device_textures.erase(name); gs_.spectrum_textures[name] = v; return; }
        }
    }
    // get_texture_mapping (textures/src/lib.rs:44-67): "uv" travels in the texture's own parameters, the other three are attached afterwards
    auto mapping_ok = [&](const std::string& mp) { return mp == "uv" || mp == "spherical" || mp == "cylindrical" || mp == "planar"; };
    auto attach_mapping = [&](uint32_t id) -> bool {
        const std::string mp = p.find_one_string("mapping", "uv");
        if (mp == "uv" || !mapping_ok(mp)) return true;   // unknown names fall back to uv with a warning (issued by the caller)
        if (mp == "planar") {
            float prm[8] = {1, 0, 0, 0, 1, 0, p.find_one_float("udelta", 0.0f), p.find_one_float("vdelta", 0.0f)};
            if (const std::vector<float>* v = p.find_floats("v1")) if (v->size() >= 3) for (int k = 0; k < 3; k++) prm[k] = (*v)[(size_t)k];
            if (const std::vector<float>* v = p.find_floats("v2")) if (v->size() >= 3) for (int k = 0; k < 3; k++) prm[3 + k] = (*v)[(size_t)k];
            return check(ABI(pbrt_hip_set_texture_mapping(scene_, id, 3, prm)), "set_texture_mapping");
        }
        return check(ABI(pbrt_hip_set_texture_mapping(scene_, id, mp == "spherical" ? 1 : 2, ctm_.mi)), "set_texture_mapping");  // tex2world.inverse()
    };
    auto forget = [&]() { gs_.unsupported_textures.erase(name); gs_.float_textures.erase(name); gs_.spectrum_textures.erase(name); gs_.device_textures.erase(name); };
    if (tex_class == "imagemap") {  // ImageTexture::from (textures/src/imagemap.rs:117-160)
        forget();
        const std::string mapping = p.find_one_string("mapping", "uv");
        if (!mapping_ok(mapping)) warn("Error 2D texture mapping '" + mapping + "' unknown");
        std::string file = p.find_one_string("filename", "");
        if (file.empty()) { if (error.empty()) error = "imagemap path not specified."; return; }
        if (file[0] != '/' && !scene_dir.empty()) file = scene_dir + "/" + file;
        const bool trilinear = p.find_one_bool("trilinear", false);
        const float max_aniso = p.find_one_float("maxanisotropy", 8.0f), scale = p.find_one_float("scale", 1.0f);
        const std::string wrap_s = p.find_one_string("wrap", "repeat");
        const int wrap = wrap_s == "black" ? 1 : (wrap_s == "clamp" ? 2 : 0);
        auto ends = [&](const char* e) { const size_t n = std::strlen(e); return file.size() >= n && file.compare(file.size() - n, n, e) == 0; };
        const bool gamma = p.find_one_bool("gamma", ends(".tga") || ends(".png"));
        char key[64];
        std::snprintf(key, sizeof key, "|%d|%d|%d|%a|%d|%a", is_float ? 1 : 0, trilinear ? 0 : 1, wrap, (double)scale, gamma ? 1 : 0, (double)max_aniso);
        const std::string ck = file + key;
        uint32_t mip = 0;
        auto it = mipmap_cache_.find(ck);
        if (it != mipmap_cache_.end()) mip = it->second;
        else {
            std::vector<float> rgb; int w = 0, h = 0; std::string err;
            if (!read_image(file, rgb, w, h, err)) { if (error.empty()) error = "Unable to load MIPMap: Error reading texture " + file + ", " + err; return; }
            if (!check(ABI(pbrt_hip_add_mipmap(scene_, w, h, rgb.data(), is_float ? 1 : 0, scale, gamma ? 1 : 0, trilinear ? 0 : 1, wrap, max_aniso, &mip)), "add_mipmap")) return;
            mipmap_cache_[ck] = mip;
        }
        uint32_t id = 0;
        if (!check(ABI(pbrt_hip_add_texture_imagemap(scene_, mip, p.find_one_float("uscale", 1.0f), p.find_one_float("vscale", 1.0f), p.find_one_float("udelta", 0.0f),
                                                     p.find_one_float("vdelta", 0.0f), &id)), "add_texture_imagemap")) return;
        if (!attach_mapping(id)) return;
        gs_.device_textures[name] = GraphicsState::DeviceTexture{is_float, id};
        return;
    }
    if (tex_class == "scale" || tex_class == "mix") {  // an operand is evaluated per hit: the whole tree goes to the library
        bool ok = true;
        auto operand = [&](const char* pn, bool want_float, std::array<float, 3> dflt) -> uint32_t {
            const std::string tn = p.find_one_texture(pn);
            if (!tn.empty()) {
                auto dt = gs_.device_textures.find(tn);
                if (dt != gs_.device_textures.end() && dt->second.is_float == want_float) return dt->second.id;
                std::array<float, 3> v = dflt; bool found = false;
                if (want_float) { auto f = gs_.float_textures.find(tn); if (f != gs_.float_textures.end()) { v = {f->second, f->second, f->second}; found = true; } }
                else { auto sp = gs_.spectrum_textures.find(tn); if (sp != gs_.spectrum_textures.end()) { v = sp->second; found = true; } }
                if (!found) { ok = false; return 0u; }
                dflt = v;
            } else if (want_float) { const float f = p.find_one_float(pn, dflt[0]); dflt = {f, f, f}; }
            else dflt = p.find_one_rgb(pn, dflt);
            uint32_t id = 0;
            if (!check(ABI(pbrt_hip_add_texture_constant(scene_, dflt.data(), &id)), "add_texture_constant")) ok = false;
            return id;
        };
        const float d1 = tex_class == "scale" ? 1.0f : 0.0f;
        const uint32_t t1 = operand("tex1", is_float, {d1, d1, d1}), t2 = operand("tex2", is_float, {1.0f, 1.0f, 1.0f});
        uint32_t id = 0;
        if (ok && tex_class == "scale") ok = check(ABI(pbrt_hip_add_texture_scale(scene_, t1, t2, &id)), "add_texture_scale");
        else if (ok) { const uint32_t amt = operand("amount", true, {0.5f, 0.5f, 0.5f}); if (ok) ok = check(ABI(pbrt_hip_add_texture_mix(scene_, t1, t2, amt, &id)), "add_texture_mix"); }
        if (ok) { forget(); gs_.device_textures[name] = GraphicsState::DeviceTexture{is_float, id}; return; }
    }
    if ((tex_class == "checkerboard" && p.find_one_int("dimension", 2) != 3) || tex_class == "uv" || tex_class == "bilerp" || tex_class == "dots") {  // 2D procedural textures (textures/src/*.rs)
        const std::string mapping = p.find_one_string("mapping", "uv");
        if (!mapping_ok(mapping)) warn("Error 2D texture mapping '" + mapping + "' unknown");
        if (tex_class == "checkerboard" && p.find_one_int("dimension", 2) != 2 && p.find_one_int("dimension", 2) != 3) {
            if (error.empty()) error = "Texture \"" + name + "\": " + std::to_string(p.find_one_int("dimension", 2)) + " dimensional checkerboard texture not supported";
            return;
        }
        if (tex_class == "uv" && is_float) { warn("Unable to create float texture 'uv'."); return; }   // textures/src/lib.rs: only a spectrum variant exists
        const float su = p.find_one_float("uscale", 1.0f), sv = p.find_one_float("vscale", 1.0f), du = p.find_one_float("udelta", 0.0f), dv = p.find_one_float("vdelta", 0.0f);
        bool ok = true;
        auto operand = [&](const char* pn, float dflt) -> uint32_t {  // a texture reference of this texture's own type, or a constant
            std::array<float, 3> v = {dflt, dflt, dflt};
            const std::string tn = p.find_one_texture(pn);
            if (!tn.empty()) {
                auto dt = gs_.device_textures.find(tn);
                if (dt != gs_.device_textures.end() && dt->second.is_float == is_float) return dt->second.id;
                bool found = false;
                if (is_float) { auto f = gs_.float_textures.find(tn); if (f != gs_.float_textures.end()) { v = {f->second, f->second, f->second}; found = true; } }
                else { auto sp = gs_.spectrum_textures.find(tn); if (sp != gs_.spectrum_textures.end()) { v = sp->second; found = true; } }
                if (!found) { ok = false; if (error.empty()) error = "Texture \"" + name + "\": operand '" + tn + "' is not a texture the library evaluates"; return 0u; }
            } else if (is_float) { const float f = p.find_one_float(pn, dflt); v = {f, f, f}; }
            else v = p.find_one_rgb(pn, v);
            uint32_t id = 0;
            if (!check(ABI(pbrt_hip_add_texture_constant(scene_, v.data(), &id)), "add_texture_constant")) ok = false;
            return id;
        };
        uint32_t id = 0;
        if (tex_class == "checkerboard") {
            const uint32_t t1 = operand("tex1", 1.0f), t2 = operand("tex2", 0.0f);
            std::string aa = p.find_one_string("aamode", "closedform");
            if (aa != "none" && aa != "closedform") { warn("Antialiasing mode '" + aa + "' not understood by Checkerboard2DTexture; using 'closedform'"); aa = "closedform"; }
            if (ok) ok = check(ABI(pbrt_hip_add_texture_checkerboard(scene_, t1, t2, su, sv, du, dv, aa == "none" ? 0 : 1, &id)), "add_texture_checkerboard");
        } else if (tex_class == "dots") {
            // Quirk B13 (textures/src/dots.rs:61-66): the reference's constructor from parameters hands (inside, outside) to DotsTexture::new(outside_dot, inside_dot), so
            // a scene file's "inside" value is what shows OUTSIDE the dots and "outside" fills them — its own render of scenes/shapes/triangles-alpha-mask.pbrt (an opaque
            // cube with holes where `"float inside" 1 "float outside" 0` asks for the opposite) confirms it.  The C ABI keeps DotsTexture's meaning; the swap lives here.
            const uint32_t param_inside = operand("inside", 1.0f), param_outside = operand("outside", 0.0f);
            if (ok) ok = check(ABI(pbrt_hip_add_texture_dots(scene_, /*inside_dot=*/param_outside, /*outside_dot=*/param_inside, su, sv, du, dv, &id)), "add_texture_dots");
        } else if (tex_class == "uv") ok = check(ABI(pbrt_hip_add_texture_uv(scene_, su, sv, du, dv, &id)), "add_texture_uv");
        else {
            auto corner = [&](const char* pn, float d) { std::array<float, 3> v = {d, d, d}; if (is_float) { const float f = p.find_one_float(pn, d); v = {f, f, f}; } else v = p.find_one_rgb(pn, v); return v; };
            const std::array<float, 3> v00 = corner("v00", 0.0f), v01 = corner("v01", 1.0f), v10 = corner("v10", 0.0f), v11 = corner("v11", 1.0f);
            ok = check(ABI(pbrt_hip_add_texture_bilerp(scene_, v00.data(), v01.data(), v10.data(), v11.data(), su, sv, du, dv, &id)), "add_texture_bilerp");
        }
        if (ok) ok = attach_mapping(id);
        if (ok) { forget(); gs_.device_textures[name] = GraphicsState::DeviceTexture{is_float, id}; }
        return;
    }
    const bool checker3d = tex_class == "checkerboard" && p.find_one_int("dimension", 2) == 3;
    if (tex_class == "fbm" || tex_class == "wrinkled" || tex_class == "windy" || tex_class == "marble" || checker3d) {  // 3D procedural textures over IdentityMapping3D
        // the reference builds IdentityMapping3D from tex2world = the CTM at the Texture directive (textures/src/fbm.rs:63-65) and applies it as is
        if (tex_class == "marble" && is_float) { warn("Unable to create float texture 'marble'."); return; }
        const float omega = p.find_one_float("roughness", 0.5f); const int octaves = p.find_one_int("octaves", 8);
        uint32_t id = 0; bool ok = true;
        if (tex_class == "fbm") ok = check(ABI(pbrt_hip_add_texture_fbm(scene_, ctm_.m, omega, octaves, &id)), "add_texture_fbm");
        else if (tex_class == "wrinkled") ok = check(ABI(pbrt_hip_add_texture_wrinkled(scene_, ctm_.m, omega, octaves, &id)), "add_texture_wrinkled");
        else if (tex_class == "windy") ok = check(ABI(pbrt_hip_add_texture_windy(scene_, ctm_.m, &id)), "add_texture_windy");
        else if (tex_class == "marble") ok = check(ABI(pbrt_hip_add_texture_marble(scene_, ctm_.m, omega, octaves, p.find_one_float("scale", 1.0f), p.find_one_float("variation", 0.2f), &id)), "add_texture_marble");
        else {
            auto operand = [&](const char* pn, float dflt) -> uint32_t {
                std::array<float, 3> v = {dflt, dflt, dflt};
                const std::string tn = p.find_one_texture(pn);
                if (!tn.empty()) {
                    auto dt = gs_.device_textures.find(tn);
                    if (dt != gs_.device_textures.end() && dt->second.is_float == is_float) return dt->second.id;
                    bool found = false;
                    if (is_float) { auto f = gs_.float_textures.find(tn); if (f != gs_.float_textures.end()) { v = {f->second, f->second, f->second}; found = true; } }
                    else { auto sp = gs_.spectrum_textures.find(tn); if (sp != gs_.spectrum_textures.end()) { v = sp->second; found = true; } }
                    if (!found) { ok = false; if (error.empty()) error = "Texture \"" + name + "\": operand '" + tn + "' is not a texture the library evaluates"; return 0u; }
                } else if (is_float) { const float f = p.find_one_float(pn, dflt); v = {f, f, f}; }
                else v = p.find_one_rgb(pn, v);
                uint32_t cid = 0;
                if (!check(ABI(pbrt_hip_add_texture_constant(scene_, v.data(), &cid)), "add_texture_constant")) ok = false;
                return cid;
            };
            const uint32_t t1 = operand("tex1", 1.0f), t2 = operand("tex2", 0.0f);
            if (ok) ok = check(ABI(pbrt_hip_add_texture_checkerboard3d(scene_, t1, t2, ctm_.m, &id)), "add_texture_checkerboard3d");
        }
        if (ok) { forget(); gs_.device_textures[name] = GraphicsState::DeviceTexture{is_float, id}; }
        return;
    }
    if (tex_class != "constant") {  // ptex: not evaluated by the library
        forget();
        gs_.unsupported_textures[name] = tex_class;
        return;
    }
    forget();
    if (is_float) gs_.float_textures[name] = p.find_one_float("value", 1.0f);
    else gs_.spectrum_textures[name] = p.find_one_rgb("value", {1.0f, 1.0f, 1.0f});
}
void Api::pbrt_material(const std::string& name, const ParamSet& p) { if (!verify_world("Material")) return; gs_.material.type = name; gs_.material.params = p; }
void Api::pbrt_make_named_material(const std::string& name, const ParamSet& p) {
    if (!verify_world("MakeNamedMaterial")) return;
    MaterialDesc m; m.type = p.find_one_string("type", ""); m.params = p;
    if (m.type.empty()) { warn("No parameter string \"type\" found in MakeNamedMaterial"); return; }
    gs_.named_materials[name] = m;
}
void Api::pbrt_named_material(const std::string& name) {
    if (!verify_world("NamedMaterial")) return;
    auto it = gs_.named_materials.find(name);
    if (it == gs_.named_materials.end()) { warn("NamedMaterial \"" + name + "\" unknown."); return; }
    gs_.material = it->second;
}
// ObjectBegin / ObjectEnd / ObjectInstance (api/src/lib.rs:911-1000)
void Api::pbrt_object_begin(const std::string& name) {
    if (!verify_world("ObjectBegin")) return;
    pbrt_attribute_begin();
    if (!current_object_.empty()) { warn("ObjectBegin called inside of an instance definition."); return; }
    uint32_t id = (uint32_t)objects_.size();
    if (!check(ABI(pbrt_hip_object_begin(scene_, &id)), "object_begin")) return;
    objects_[name] = id; object_tris_[name] = 0;
    current_object_ = name;
}
void Api::pbrt_object_end() {
    if (!verify_world("ObjectEnd")) return;
    if (current_object_.empty()) warn("ObjectEnd called outside of instance definition.");
    else { check(ABI(pbrt_hip_object_end(scene_)), "object_end"); current_object_.clear(); }
    pbrt_attribute_end();
}
void Api::pbrt_object_instance(const std::string& name) {
    if (!verify_world("ObjectInstance") || !error.empty()) return;
    if (!current_object_.empty()) { warn("ObjectInstance can't be called inside of instance definition."); return; }
    auto it = objects_.find(name);
    if (it == objects_.end()) { warn("Unable to find object instance named '" + name + "'"); return; }
    if (object_tris_[name] == 0) return;  // empty object: nothing is added (lib.rs:949-951)
    if (!check(ABI(pbrt_hip_add_instance(scene_, it->second, ctm_.m, ctm_.mi)), "add_instance")) return;
    n_instances_++; n_tris_ += object_tris_[name];
}
void Api::pbrt_reverse_orientation() { if (!verify_world("ReverseOrientation")) return; gs_.reverse_orientation = !gs_.reverse_orientation; }

static std::array<float, 3> mul3(std::array<float, 3> a, std::array<float, 3> b) { return {a[0] * b[0], a[1] * b[1], a[2] * b[2]}; }

void Api::pbrt_light_source(const std::string& name, const ParamSet& p, const std::string& scene_dir) {
    if (!verify_world("LightSource") || !error.empty() || (!scene_ && !check_only_)) return;
    for (auto& u : p.unsupported) { error = "LightSource \"" + name + "\": parameter '" + u + "' has a spectral type this host cannot evaluate"; return; }
    const std::array<float, 3> one = {1.0f, 1.0f, 1.0f};
    const std::array<float, 3> sc = p.find_one_rgb("scale", one);
    if (name == "infinite" || name == "exinfinite") {
        auto L = mul3(p.find_one_rgb("L", one), sc);
        std::string map = p.find_one_string("mapname", "");
        std::vector<float> rgb; int w = 0, h = 0;
        if (!map.empty()) {  // InfiniteAreaLight::new: a map that cannot be read leaves a constant light, with a warning (infinite.rs:63-77)
            if (map[0] != '/' && !scene_dir.empty()) map = scene_dir + "/" + map;
            std::string err;
            if (!read_image(map, rgb, w, h, err)) {
                rgb.clear();
                // a file the reference could read but this host cannot decode must not silently become a constant sky
                if (err.find("not decoded by this host") != std::string::npos) { if (error.empty()) error = "LightSource \"infinite\" 'mapname' " + map + ": " + err; return; }
                warn("Problem reading file '" + map + "'. " + err);
            }
        }
        if (!rgb.empty()) { if (check(ABI(pbrt_hip_add_light_infinite_map(scene_, L.data(), w, h, rgb.data(), ctm_.m, ctm_.mi)), "add_light_infinite_map")) n_lights_++; }
        else if (check(ABI(pbrt_hip_add_light_infinite(scene_, L.data(), ctm_.m, ctm_.mi)), "add_light_infinite")) n_lights_++;
    } else if (name == "distant") {
        auto L = mul3(p.find_one_rgb("L", one), sc);
        auto from = p.find_one_rgb("from", {0.0f, 0.0f, 0.0f}), to = p.find_one_rgb("to", {0.0f, 0.0f, 1.0f});
        float w[3];
        pbrt_hip_host_distant_direction(ctm_.m, from.data(), to.data(), w);
        if (check(ABI(pbrt_hip_add_light_distant(scene_, L.data(), w)), "add_light_distant")) n_lights_++;
    } else if (name == "point") {
        auto I = mul3(p.find_one_rgb("I", one), sc);
        auto from = p.find_one_rgb("from", {0.0f, 0.0f, 0.0f});
        float pw[3];
        pbrt_hip_host_point_position(ctm_.m, ctm_.mi, from.data(), pw);
        if (check(ABI(pbrt_hip_add_light_point(scene_, I.data(), pw)), "add_light_point")) n_lights_++;
    } else if (name == "spot") {
        auto I = mul3(p.find_one_rgb("I", one), sc);
        auto from = p.find_one_rgb("from", {0.0f, 0.0f, 0.0f}), to = p.find_one_rgb("to", {0.0f, 0.0f, 1.0f});
        float l2w[16], w2l[16], cs[2];
        pbrt_hip_host_spot(ctm_.m, ctm_.mi, from.data(), to.data(), p.find_one_float("coneangle", 30.0f), p.find_one_float("conedeltaangle", 5.0f), l2w, w2l, cs);
        if (check(ABI(pbrt_hip_add_light_spot(scene_, I.data(), l2w, w2l, cs[0], cs[1])), "add_light_spot")) n_lights_++;
    } else if (name == "projection" || name == "goniometric") {   // projection.rs:275-296, goniometric.rs:220-237: light_to_world = the CTM
        auto I = mul3(p.find_one_rgb("I", one), sc);
        std::string map = p.find_one_string("mapname", "");
        std::vector<float> rgb; int w = 0, h = 0;
        if (map.empty()) warn(name == "projection" ? "No projection image texture provided." : "No goniophotometric image texture provided.");
        else {   // an image that cannot be read leaves the light without one, with a warning (projection.rs:83-86, goniometric.rs:60-63)
            if (map[0] != '/' && !scene_dir.empty()) map = scene_dir + "/" + map;
            std::string err;
            if (!read_image(map, rgb, w, h, err)) {
                rgb.clear();
                if (err.find("not decoded by this host") != std::string::npos) { if (error.empty()) error = "LightSource \"" + name + "\" 'mapname' " + map + ": " + err; return; }
                warn("Problem reading file '" + map + "'. " + err);
            }
        }
        const float* img = rgb.empty() ? nullptr : rgb.data();
        if (name == "projection") { if (check(ABI(pbrt_hip_add_light_projection(scene_, I.data(), ctm_.m, ctm_.mi, p.find_one_float("fov", 45.0f), w, h, img)), "add_light_projection")) n_lights_++; }
        else if (check(ABI(pbrt_hip_add_light_goniometric(scene_, I.data(), ctm_.m, ctm_.mi, w, h, img)), "add_light_goniometric")) n_lights_++;
    } else {
        error = "LightSource \"" + name + "\" is outside the hot-path scope (supported: infinite, distant, point, spot, projection, goniometric and diffuse area lights)";
    }
}
void Api::pbrt_area_light_source(const std::string& name, const ParamSet& p) { if (!verify_world("AreaLightSource")) return; gs_.area_light = name; gs_.area_light_params = p; }

// MatteMaterial From<&TextureParams> (materials/src/matte.rs:95-110) with TextureParams's lookup order: shape parameters
// first, then the material's own (core/src/paramset/texture_params.rs).
uint32_t Api::material_id_for(const MaterialDesc& m) {
    std::array<float, 3> kd = {0.5f, 0.5f, 0.5f};
    float sigma = 0.0f;
    int64_t ftex_param[4] = {-1, -1, -1, -1};     // [sigma, uroughness, vroughness, index]: the same for float parameters
    int64_t tex_param[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // [Kd, Ks, Kr, Kt, opacity, amount, eta, k, reflect, transmit]: the parameter names a texture the library evaluates per hit
    auto spectrum_tex = [&](const std::string& pname, std::array<float, 3> d) {
        std::string tn = m.params.find_one_texture(pname);
        if (!tn.empty()) {
            auto dt = gs_.device_textures.find(tn);
            if (dt != gs_.device_textures.end()) {
                // a texture the library evaluates per hit: the material is created with a white placeholder and the texture attached afterwards
                const int param = pname == "Kd" ? 0 : (pname == "Ks" ? 1 : (pname == "Kr" ? 2 : (pname == "Kt" ? 3 : (pname == "opacity" ? 4 : (pname == "amount" ? 5 : (pname == "eta" ? 6 : (pname == "k" ? 7 : (pname == "reflect" ? 8 : (pname == "transmit" ? 9 : -1)))))))));
                const bool takes = (m.type == "matte" && param == 0) || ((m.type == "plastic" || m.type == "substrate") && (param == 0 || param == 1)) || (m.type == "mirror" && param == 2) ||
                                   (m.type == "glass" && (param == 2 || param == 3)) || (m.type == "uber" && param >= 0 && param <= 4) || (m.type == "translucent" && (param == 0 || param == 1 || param == 8 || param == 9)) ||
                                   (m.type == "mix" && param == 5) || (m.type == "metal" && (param == 6 || param == 7));
                if (!takes || dt->second.is_float) {
                    if (error.empty()) error = "texture '" + tn + "' on parameter '" + pname + "' of Material \"" + m.type + "\": per-hit spectrum textures are wired to matte Kd, plastic Kd / Ks, mirror Kr, substrate Kd / Ks, glass Kr / Kt, uber Kd / Ks / Kr / Kt / opacity, translucent Kd / Ks / reflect / transmit, mix amount and metal eta / k";
                    return m.params.find_one_rgb(pname, d);
                }
                tex_param[param] = (int64_t)dt->second.id;
                return std::array<float, 3>{1.0f, 1.0f, 1.0f};
            }
            auto it = gs_.spectrum_textures.find(tn);
            if (it != gs_.spectrum_textures.end()) return it->second;
            auto un = gs_.unsupported_textures.find(tn);
            if (un != gs_.unsupported_textures.end() && error.empty())
                error = "texture '" + tn + "' of class '" + un->second + "' is not evaluated by the library yet (constant, scale, mix and imagemap are)";
            else if (error.empty()) warn("Couldn't find spectrum texture named '" + tn + "' for parameter '" + pname + "'");
        }
        return m.params.find_one_rgb(pname, d);
    };
    auto float_tex = [&](const std::string& pname, float d) {
        std::string tn = m.params.find_one_texture(pname);
        if (!tn.empty()) {
            auto dtf = gs_.device_textures.find(tn);
            if (dtf != gs_.device_textures.end()) {
                // a float texture the library evaluates per hit: sigma (matte) or the microfacet roughness (plastic, uber, substrate, metal, translucent)
                const int fp = pname == "sigma" ? 0 : ((pname == "uroughness" || pname == "roughness") ? 1 : (pname == "vroughness" ? 2 : (pname == "index" ? 3 : -1)));
                const bool takes = dtf->second.is_float && ((m.type == "matte" && fp == 0) || ((m.type == "plastic" || m.type == "uber" || m.type == "substrate" || m.type == "metal" || m.type == "translucent") && (fp == 1 || fp == 2)) ||
                                                            (m.type == "glass" && (fp == 1 || fp == 2) && pname != "roughness") || ((m.type == "glass" || m.type == "uber") && fp == 3));
                if (!takes) {
                    if (error.empty()) error = "texture '" + tn + "' on parameter '" + pname + "' of Material \"" + m.type + "\": per-hit float textures are wired to matte sigma and to the roughness of plastic / uber / substrate / metal / translucent / glass";
                    return m.params.find_one_float(pname, d);
                }
                if (pname == "roughness") { if (ftex_param[1] < 0) ftex_param[1] = (int64_t)dtf->second.id; if (ftex_param[2] < 0) ftex_param[2] = (int64_t)dtf->second.id; }
                else ftex_param[fp] = (int64_t)dtf->second.id;
                return d;   // placeholder: the texture replaces it
            }
            auto it = gs_.float_textures.find(tn);
            if (it != gs_.float_textures.end()) return it->second;
            auto un = gs_.unsupported_textures.find(tn);
            if (un != gs_.unsupported_textures.end() && error.empty())
                error = "texture '" + tn + "' of class '" + un->second + "' is not evaluated by the library yet (constant, scale, mix and imagemap are)";
            else if (error.empty()) warn("Couldn't find float texture named '" + tn + "' for parameter '" + pname + "'");
        }
        return m.params.find_one_float(pname, d);
    };
    // `bumpmap`: get_float_texture_or_none (every material's From<&TextureParams>): a float texture by name, never a literal
    int64_t bump_tex = -1;
    {
        const std::string bn = m.params.find_one_texture("bumpmap");
        if (!bn.empty()) {
            auto dt = gs_.device_textures.find(bn);
            if (dt != gs_.device_textures.end() && dt->second.is_float) bump_tex = (int64_t)dt->second.id;
            else if (gs_.float_textures.count(bn)) {
                const float v = gs_.float_textures[bn]; const float c3_[3] = {v, v, v};
                uint32_t id = 0;
                if (!check(ABI(pbrt_hip_add_texture_constant(scene_, c3_, &id)), "add_texture_constant")) return 0;
                bump_tex = (int64_t)id;
            } else if (gs_.unsupported_textures.count(bn)) { if (error.empty()) error = "'bumpmap' texture '" + bn + "' of class '" + gs_.unsupported_textures[bn] + "' is not evaluated by the library"; }
            else warn("Couldn't find float texture named '" + bn + "' for parameter 'bumpmap'");
            if (bump_tex >= 0 && (m.type == "mix" || m.type == "none" || m.type.empty())) bump_tex = -1;   // MixMaterial and "none" have no bump map
        }
    }
    const bool remap = m.params.find_one_bool("remaproughness", true);
    // every float that defines the material, in order, is the cache key
    std::vector<float> kv;
    auto put3 = [&](const std::array<float, 3>& a) { kv.insert(kv.end(), a.begin(), a.end()); };
    const std::array<float, 3> zero = {0.0f, 0.0f, 0.0f}, one = {1.0f, 1.0f, 1.0f}, quarter = {0.25f, 0.25f, 0.25f};
    std::array<float, 3> a3 = zero, b3 = zero, c3 = zero, d3 = zero, e3 = zero;
    float f0 = 0, f1 = 0, f2 = 0;
    const std::string& t = m.type;
    // Quirk B14 (glass.rs:158-161, uber.rs:201-204): the reference asks `tp.get_float_texture("eta")`, which is a lookup in the table of NAMED float textures — one
    // declared as `Texture "eta" "float" ...` — and not a parameter look-up: the `"float eta" 2` that pbrt-v3 scenes (and the reference's own depth-of-field.pbrt) write is never
    // read.  Without such a texture the index of refraction is the "index" parameter, default 1.5.  Its render of depth-of-field.pbrt shows eta 1.5 spheres, pixel for pixel.
    auto eta_of = [&]() {
        auto named = gs_.float_textures.find("eta");
        if (named != gs_.float_textures.end()) return named->second;   // a constant float texture that happens to be called "eta"
        auto dev = gs_.device_textures.find("eta");
        if (dev != gs_.device_textures.end() && dev->second.is_float) { ftex_param[3] = (int64_t)dev->second.id; return 1.5f; }   // the float texture NAMED "eta": the index of refraction of every hit (glass.rs:158-161)
        if (dev != gs_.device_textures.end() || gs_.unsupported_textures.count("eta")) {
            if (error.empty()) error = "Material \"" + m.type + "\": the texture named \"eta\" would set the index of refraction per hit (glass.rs:158), but it is not a float texture the library evaluates";
            return 1.5f;
        }
        if (m.params.floats.count("eta") || !m.params.find_one_texture("eta").empty())
            warn("Material \"" + m.type + "\": the reference does not read the parameter \"eta\" (it looks up a float texture NAMED \"eta\", glass.rs:158 / uber.rs:201); using \"index\"");
        return float_tex("index", 1.5f);
    };
    auto uv_rough = [&](float dflt, float& u, float& v) {  // `uroughness` / `vroughness` fall back to `roughness` (metal.rs:69-76, uber.rs:148-155)
        const float r = float_tex("roughness", dflt);
        const bool hu = m.params.floats.count("uroughness") || !m.params.find_one_texture("uroughness").empty();
        const bool hv = m.params.floats.count("vroughness") || !m.params.find_one_texture("vroughness").empty();
        u = hu ? float_tex("uroughness", r) : r; v = hv ? float_tex("vroughness", r) : r;
    };
    if (t == "none" || t.empty()) { /* no BSDF: graphics_state.rs make_material returns None */ }
    else if (t == "matte") {
        a3 = spectrum_tex("Kd", kd); f0 = float_tex("sigma", sigma); put3(a3); kv.push_back(f0);
    }
    else if (t == "mirror") { a3 = spectrum_tex("Kr", {0.9f, 0.9f, 0.9f}); put3(a3); }
    else if (t == "plastic") { a3 = spectrum_tex("Kd", quarter); b3 = spectrum_tex("Ks", quarter); f0 = float_tex("roughness", 0.1f); put3(a3); put3(b3); kv.push_back(f0); }
    else if (t == "glass") {
        a3 = spectrum_tex("Kr", one); b3 = spectrum_tex("Kt", one); f0 = float_tex("uroughness", 0.0f); f1 = float_tex("vroughness", 0.0f); f2 = eta_of();
        put3(a3); put3(b3); kv.push_back(f0); kv.push_back(f1); kv.push_back(f2);
    } else if (t == "metal") {
        std::array<float, 3> cu_n, cu_k;   // the defaults are copper's measured n and k (metal.rs:136-147)
        pbrt_hip_host_copper_rgb(cu_n.data(), cu_k.data());
        a3 = spectrum_tex("eta", cu_n); b3 = spectrum_tex("k", cu_k); uv_rough(0.01f, f0, f1);
        put3(a3); put3(b3); kv.push_back(f0); kv.push_back(f1);
    } else if (t == "uber") {
        a3 = spectrum_tex("Kd", quarter); b3 = spectrum_tex("Ks", quarter); c3 = spectrum_tex("Kr", zero); d3 = spectrum_tex("Kt", zero); e3 = spectrum_tex("opacity", one);
        uv_rough(0.1f, f0, f1); f2 = eta_of();
        put3(a3); put3(b3); put3(c3); put3(d3); put3(e3); kv.push_back(f0); kv.push_back(f1); kv.push_back(f2);
    } else if (t == "substrate") {
        a3 = spectrum_tex("Kd", {0.5f, 0.5f, 0.5f}); b3 = spectrum_tex("Ks", {0.5f, 0.5f, 0.5f}); f0 = float_tex("uroughness", 0.1f); f1 = float_tex("vroughness", 0.1f);
        put3(a3); put3(b3); kv.push_back(f0); kv.push_back(f1);
    } else if (t == "translucent") {
        a3 = spectrum_tex("Kd", quarter); b3 = spectrum_tex("Ks", quarter); c3 = spectrum_tex("reflect", {0.5f, 0.5f, 0.5f}); d3 = spectrum_tex("transmit", {0.5f, 0.5f, 0.5f});
        f0 = float_tex("roughness", 0.1f);
        put3(a3); put3(b3); put3(c3); put3(d3); kv.push_back(f0);
    } else if (t == "mix") {  // graphics_state.rs:310-330: two named materials, an unknown name falls back to matte made from the same parameters
        a3 = spectrum_tex("amount", {0.5f, 0.5f, 0.5f}); put3(a3);
        uint32_t sub[2];
        const char* pn[2] = {"namedmaterial1", "namedmaterial2"};
        for (int k = 0; k < 2; k++) {
            const std::string nm = m.params.find_one_string(pn[k], "");
            auto it = gs_.named_materials.find(nm);
            MaterialDesc md;
            if (it != gs_.named_materials.end()) md = it->second;
            else { warn("Named material '" + nm + "' undefined. Using 'matte'."); md.type = "matte"; md.params = m.params; }
            sub[k] = material_id_for(md);
            if (!error.empty()) return 0;
            kv.push_back((float)sub[k]);
        }
        f0 = (float)sub[0]; f1 = (float)sub[1];
    } else {
        if (error.empty()) error = "Material \"" + t + "\" is outside the hot-path scope (supported: matte, mirror, plastic, glass, metal, uber, substrate, translucent, mix)";
        return 0;
    }
    if (!error.empty()) return 0;
    std::string key = t + (remap ? ":r" : ":n");
    for (float v : kv) { uint32_t u; std::memcpy(&u, &v, 4); char b[12]; std::snprintf(b, sizeof b, ":%08x", u); key += b; }
    if (bump_tex >= 0) key += "|bump=" + std::to_string(bump_tex);
    for (int k = 0; k < 4; k++) if (ftex_param[k] >= 0) key += "|ftex" + std::to_string(k) + "=" + std::to_string(ftex_param[k]);
    for (int k = 0; k < 10; k++) if (tex_param[k] >= 0) key += "|tex" + std::to_string(k) + "=" + std::to_string(tex_param[k]);
    auto it = material_cache_.find(key);
    if (it != material_cache_.end()) return it->second;
    uint32_t id = 0;
    int rc;
    if (t == "none" || t.empty()) rc = ABI(pbrt_hip_add_material_none(scene_, &id));
    else if (t == "matte") rc = ABI(pbrt_hip_add_material_matte(scene_, a3.data(), f0, &id));
    else if (t == "mirror") rc = ABI(pbrt_hip_add_material_mirror(scene_, a3.data(), &id));
    else if (t == "plastic") rc = ABI(pbrt_hip_add_material_plastic(scene_, a3.data(), b3.data(), f0, remap ? 1 : 0, &id));
    else if (t == "glass") rc = ABI(pbrt_hip_add_material_glass(scene_, a3.data(), b3.data(), f0, f1, f2, remap ? 1 : 0, &id));
    else if (t == "metal") rc = ABI(pbrt_hip_add_material_metal(scene_, a3.data(), b3.data(), f0, f1, remap ? 1 : 0, &id));
    else if (t == "substrate") rc = ABI(pbrt_hip_add_material_substrate(scene_, a3.data(), b3.data(), f0, f1, remap ? 1 : 0, &id));
    else if (t == "translucent") rc = ABI(pbrt_hip_add_material_translucent(scene_, a3.data(), b3.data(), c3.data(), d3.data(), f0, remap ? 1 : 0, &id));
    else if (t == "mix") rc = ABI(pbrt_hip_add_material_mix(scene_, (uint32_t)f0, (uint32_t)f1, a3.data(), &id));
    else rc = ABI(pbrt_hip_add_material_uber(scene_, a3.data(), b3.data(), c3.data(), d3.data(), e3.data(), f0, f1, f2, remap ? 1 : 0, &id));
    if (!check(rc, "add_material")) return 0;
    // colours first, then the scalars, then the structural parameters (opacity / amount rebuild or annotate the lobe list the others have filled in)
    for (int k = 0; k < 4; k++)
        if (tex_param[k] >= 0 && !check(ABI(pbrt_hip_set_material_texture(scene_, id, k, (uint32_t)tex_param[k])), "set_material_texture")) return 0;
    for (int k = 0; k < 3; k++)
        if (ftex_param[k] >= 0 && !check(ABI(pbrt_hip_set_material_float_texture(scene_, id, k, (uint32_t)ftex_param[k])), "set_material_float_texture")) return 0;
    for (int k = 4; k < 8; k++)
        if (tex_param[k] >= 0 && !check(ABI(pbrt_hip_set_material_texture(scene_, id, k, (uint32_t)tex_param[k])), "set_material_texture")) return 0;
    if (ftex_param[3] >= 0 && !check(ABI(pbrt_hip_set_material_float_texture(scene_, id, 3, (uint32_t)ftex_param[3])), "set_material_float_texture")) return 0;   // index: after opacity (an uber's is made per hit with it)
    for (int k = 8; k < 10; k++)   // translucent's reflect / transmit: after Kd / Ks and the roughness (the per-hit lobe list takes their textures over)
        if (tex_param[k] >= 0 && !check(ABI(pbrt_hip_set_material_texture(scene_, id, k, (uint32_t)tex_param[k])), "set_material_texture")) return 0;
    if (bump_tex >= 0 && !check(ABI(pbrt_hip_set_material_bump(scene_, id, (uint32_t)bump_tex)), "set_material_bump")) return 0;
    material_cache_[key] = id;
    return id;
}

void Api::pbrt_shape(const std::string& name, const ParamSet& p, const std::string& scene_dir) {
    if (!verify_world("Shape") || !error.empty() || (!scene_ && !check_only_)) return;
    std::vector<float> P, N, S, UV;
    std::vector<uint32_t> idx;
    if (name == "trianglemesh") {  // shapes/src/triangle.rs:184-330
        const std::vector<int>* vi = p.find_ints("indices");
        const std::vector<float>* pp = p.find_floats("P");
        if (!vi || vi->empty()) { warn("Vertex indices 'indices' not provided with triangle mesh shape"); return; }
        if (!pp || pp->empty()) { warn("Vertex positions 'P' not provided with triangle mesh shape"); return; }
        const size_t npi = pp->size() / 3;
        P.assign(pp->begin(), pp->begin() + npi * 3);
        const std::vector<float>* uv = p.find_floats("uv");
        if (!uv || uv->empty()) uv = p.find_floats("st");
        if (uv && !uv->empty()) {
            const size_t nuv = uv->size() / 2;
            if (nuv < npi) warn("Not enough of 'uv' for triangle mesh. Discarding.");
            else { if (nuv > npi) warn("More 'uv' provided than will be used for triangle mesh."); UV.assign(uv->begin(), uv->begin() + npi * 2); }
        }
        if (const std::vector<float>* s = p.find_floats("S")) {
            if (s->size() / 3 != npi) warn("Number of 'S' for triangle mesh must match 'P'."); else S = *s;
        }
        if (const std::vector<float>* n = p.find_floats("N")) {
            if (n->size() / 3 != npi) warn("Number of 'N' for triangle mesh must match 'P'."); else N = *n;
        }
        for (int v : *vi)
            if (v < 0 || (size_t)v >= npi) { warn("trianglemesh has out-of-bounds vertex index " + std::to_string(v)); return; }
        idx.assign(vi->begin(), vi->end());
        idx.resize(idx.size() / 3 * 3);
    } else if (name == "plymesh") {  // shapes/src/plymesh.rs:21-141
        std::string fn = p.find_one_string("filename", "");
        if (fn.empty()) { error = "plymesh: no 'filename' parameter"; return; }
        if (fn[0] != '/' && !scene_dir.empty()) fn = scene_dir + "/" + fn;
        PlyMesh mesh; std::string err;
        if (!read_ply(fn, mesh, err)) { error = "Unable to parse PLY file '" + fn + "'. " + err; return; }
        if (mesh.P.empty() || mesh.indices.empty()) { warn("PLY file '" + fn + "' is invalid! No face/vertex elements found!"); return; }
        P.swap(mesh.P); N.swap(mesh.N); UV.swap(mesh.UV); idx.swap(mesh.indices);
    } else {
        error = "Shape \"" + name + "\" is outside the hot-path scope (supported: trianglemesh, plymesh)";
        return;
    }
    // alpha / shadowalpha (triangle.rs:278-312): a float texture by name (constant ones fold to the constant, the others are evaluated per candidate hit) or a float
    uint32_t alpha_tex[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    auto alpha_of = [&](const char* pname) {
        std::string tn = p.find_one_texture(pname);
        if (!tn.empty()) {
            auto dt = gs_.device_textures.find(tn);
            if (dt != gs_.device_textures.end() && dt->second.is_float) { alpha_tex[std::strcmp(pname, "alpha") ? 1 : 0] = dt->second.id; return 1.0f; }
            auto it = gs_.float_textures.find(tn);
            if (it != gs_.float_textures.end()) return it->second;
            auto un = gs_.unsupported_textures.find(tn);
            if (un != gs_.unsupported_textures.end()) { if (error.empty()) error = std::string("'") + pname + "' texture '" + tn + "' of class '" + un->second + "' is outside the hot-path scope"; }
            else warn("Couldn't find float texture '" + tn + "' for '" + pname + "' parameter. Using float parameter instead.");
        }
        return p.find_one_float(pname, 1.0f);
    };
    const float alpha = alpha_of("alpha"), shadow_alpha = alpha_of("shadowalpha");
    if (!error.empty()) return;

    // material: shape parameters override the material's (TextureParams looks in the shape's set first, graphics_state.rs:147-165)
    MaterialDesc eff = gs_.material;
    static const char* kMatParams[] = {"Kd", "Ks", "Kr", "Kt", "sigma", "roughness", "uroughness", "vroughness", "eta", "index", "k", "opacity", "bumpmap", "reflect", "transmit", "amount"};
    for (const char* name : kMatParams) {
        auto f = p.floats.find(name); auto tx = p.textures.find(name);
        if (tx != p.textures.end()) { eff.params.textures[name] = tx->second; eff.params.floats.erase(name); }
        else if (f != p.floats.end()) { eff.params.floats[name] = f->second; eff.params.textures.erase(name); }
    }
    { auto b = p.bools.find("remaproughness"); if (b != p.bools.end()) eff.params.bools["remaproughness"] = b->second; }
    const uint32_t mat = material_id_for(eff);
    if (!error.empty()) return;

    // TriangleMesh::new moves the vertices to world space once (triangle.rs:93-99)
    const size_t nv = P.size() / 3;
    std::vector<float> Pw(P.size());
    pbrt_hip_host_transform_points(ctm_.m, P.data(), Pw.data(), nv);
    if (!N.empty()) { std::vector<float> t(N.size()); pbrt_hip_host_transform_normals(ctm_.mi, N.data(), t.data(), nv); N.swap(t); }
    if (!S.empty()) { std::vector<float> t(S.size()); pbrt_hip_host_transform_vectors(ctm_.m, S.data(), t.data(), nv); S.swap(t); }
    const uint32_t n_tris = (uint32_t)(idx.size() / 3);
    const uint32_t flags = (gs_.reverse_orientation ? 1u : 0u) | (pbrt_hip_host_swaps_handedness(ctm_.m) ? 2u : 0u);

    int32_t first_light = -1;
    const bool in_object = !current_object_.empty();
    if (!gs_.area_light.empty()) {  // one DiffuseAreaLight per triangle, numbered where the Shape directive stands (lib.rs:783-812)
        if (gs_.area_light != "diffuse" && gs_.area_light != "area") { error = "AreaLightSource \"" + gs_.area_light + "\" unknown"; return; }
        const ParamSet& ap = gs_.area_light_params;
        for (auto& u : ap.unsupported) { error = "AreaLightSource: parameter '" + u + "' has a spectral type this host cannot evaluate"; return; }
        auto L = mul3(ap.find_one_rgb("L", {1.0f, 1.0f, 1.0f}), ap.find_one_rgb("scale", {1.0f, 1.0f, 1.0f}));
        uint32_t id = 0;
        if (!check(ABI(pbrt_hip_add_light_diffuse_area(scene_, L.data(), ap.find_one_bool("twosided", false) ? 1 : 0, n_tris, &id)), "add_light_diffuse_area")) return;
        first_light = (int32_t)id;
        if (in_object) warn("Area lights not supported with object instancing.");   // lib.rs:877-881: the shape keeps its emission, the scene's lights do not get the light (pbrt_hip_add_mesh does the same)
        else n_lights_ += n_tris;
    }
    if (!check(ABI(pbrt_hip_add_mesh(scene_, Pw.data(), (uint32_t)nv, idx.data(), n_tris, N.empty() ? nullptr : N.data(), S.empty() ? nullptr : S.data(),
                                 UV.empty() ? nullptr : UV.data(), mat, first_light, flags, alpha, shadow_alpha)), "add_mesh")) return;
    if ((alpha_tex[0] != 0xFFFFFFFFu || alpha_tex[1] != 0xFFFFFFFFu) && !check(ABI(pbrt_hip_set_last_mesh_alpha_textures(scene_, alpha_tex[0], alpha_tex[1])), "set_last_mesh_alpha_textures")) return;
    if (current_object_.empty()) n_tris_ += n_tris; else object_tris_[current_object_] += n_tris;
}


int Api::pbrt_world_end(RenderReport& rep) {
    using clk = std::chrono::steady_clock;
    if (!scene_ && !check_only_) return PBRT_HIP_ERR_NO_DEVICE;
    if (!error.empty()) return PBRT_HIP_ERR_UNSUPPORTED;
    if (!verify_world("WorldEnd")) { error = "WorldEnd outside a world block"; return PBRT_HIP_ERR_STATE; }
    while (!gs_stack_.empty()) { warn("Missing end to AttributeBegin"); pbrt_attribute_end(); }
    while (!ctm_stack_.empty()) { warn("Missing end to TransformBegin"); ctm_stack_.pop_back(); }

    // ---- filter + film (make_filter / make_film, graphics_state.rs:600-690; film/mod.rs:420-487)
    int fkind = -1; float rad[2] = {0.5f, 0.5f}, fparams[2] = {0.0f, 0.0f};
    if (filter_name_ == "box") { fkind = 0; rad[0] = filter_p_.find_one_float("xwidth", 0.5f); rad[1] = filter_p_.find_one_float("ywidth", 0.5f); }
    else if (filter_name_ == "gaussian") { fkind = 1; rad[0] = filter_p_.find_one_float("xwidth", 2.0f); rad[1] = filter_p_.find_one_float("ywidth", 2.0f); fparams[0] = filter_p_.find_one_float("alpha", 2.0f); }
    else if (filter_name_ == "mitchell") { fkind = 2; rad[0] = filter_p_.find_one_float("xwidth", 2.0f); rad[1] = filter_p_.find_one_float("ywidth", 2.0f); fparams[0] = filter_p_.find_one_float("B", 1.0f / 3.0f); fparams[1] = filter_p_.find_one_float("C", 1.0f / 3.0f); }
    else if (filter_name_ == "sinc") { fkind = 3; rad[0] = filter_p_.find_one_float("xwidth", 4.0f); rad[1] = filter_p_.find_one_float("ywidth", 4.0f); fparams[0] = filter_p_.find_one_float("tau", 3.0f); }
    else if (filter_name_ == "triangle") { fkind = 4; rad[0] = filter_p_.find_one_float("xwidth", 2.0f); rad[1] = filter_p_.find_one_float("ywidth", 2.0f); }
    else { error = "Filter \"" + filter_name_ + "\" unknown."; return PBRT_HIP_ERR_UNSUPPORTED; }
    if (film_name_ != "image") { error = "Film \"" + film_name_ + "\" unknown."; return PBRT_HIP_ERR_UNSUPPORTED; }
    const int xres = film_p_.find_one_int("xresolution", 1280), yres = film_p_.find_one_int("yresolution", 720);
    float crop[4] = {0.0f, 1.0f, 0.0f, 1.0f};
    auto clamp01 = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
    if (const std::vector<float>* cr = film_p_.find_floats("cropwindow")) {
        if (cr->size() != 4) { error = std::to_string(cr->size()) + " values supplied for 'cropwindow'. Expected 4."; return PBRT_HIP_ERR_INVALID_ARG; }
        const float* c = cr->data();
        crop[0] = clamp01(c[0] < c[1] ? c[0] : c[1]); crop[1] = clamp01(c[0] > c[1] ? c[0] : c[1]);
        crop[2] = clamp01(c[2] < c[3] ? c[2] : c[3]); crop[3] = clamp01(c[2] > c[3] ? c[2] : c[3]);
    } else if (has_crop_override) {
        for (int i = 0; i < 4; i++) crop[i] = clamp01(crop_override[i]);
    }
    std::string filename = film_p_.find_one_string("filename", "pbrt.exr");
    if (!override_outfile.empty()) {
        if (film_p_.strings.count("filename")) warn("Output filename supplied on command line, '" + override_outfile + "' is overriding filename provided in scene description file, '" + filename + "'.");
        filename = override_outfile;
    }
    {   // write_image (image_io.rs:225-237) knows .exr, .tga, .png and .pfm; anything else is an error before rendering, not after
        size_t dot = filename.find_last_of('.'), slash = filename.find_last_of('/');
        std::string ext = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? filename.substr(dot) : "";
        for (char& c : ext) c = (char)std::tolower((unsigned char)c);
        if (!(ext == ".exr" || ext == ".tga" || ext == ".png" || ext == ".pfm")) {
            error = ext.empty() ? "Can't determine file type from suffix of filename " + filename : "Extension " + ext + " is not supported";
            return PBRT_HIP_ERR_INVALID_ARG;
        }
    }
    const float film_scale = film_p_.find_one_float("scale", 1.0f);
    const float max_lum = film_p_.find_one_float("maxsampleluminance", std::numeric_limits<float>::infinity());
    int cb[4], sb[4]; float table[256];
    if (pbrt_hip_host_film_filter(fkind, fparams, xres, yres, crop, rad, cb, table, sb) != 0) { error = "film filter set-up failed"; return PBRT_HIP_ERR_INVALID_ARG; }
    if (!check(ABI(pbrt_hip_set_film(scene_, xres, yres, cb, rad, table, film_scale, max_lum)), "set_film")) return PBRT_HIP_ERR_INVALID_ARG;

    // ---- camera (perspective_camera.rs:358-419, orthographic_camera.rs:188-244: the same parameters minus the field of view)
    if (camera_name_ != "perspective" && camera_name_ != "orthographic" && camera_name_ != "environment") { error = "Camera \"" + camera_name_ + "\" is outside the hot-path scope (supported: perspective, orthographic, environment)"; return PBRT_HIP_ERR_UNSUPPORTED; }
    float shutter_open = camera_p_.find_one_float("shutteropen", 0.0f), shutter_close = camera_p_.find_one_float("shutterclose", 1.0f);
    if (shutter_close < shutter_open) { warn("Shutter close time < shutter open. Swapping them."); std::swap(shutter_open, shutter_close); }
    const float lens_radius = camera_p_.find_one_float("lensradius", 0.0f), focal_distance = camera_p_.find_one_float("focaldistance", 1e6f);
    const float frame = camera_p_.find_one_float("frameaspectratio", (float)xres / (float)yres);
    float screen[4];
    if (frame > 1.0f) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.0f; screen[3] = 1.0f; }
    else { screen[0] = -1.0f; screen[1] = 1.0f; screen[2] = -1.0f / frame; screen[3] = 1.0f / frame; }
    if (const std::vector<float>* sw = camera_p_.find_floats("screenwindow")) {
        if (sw->size() == 4) for (int i = 0; i < 4; i++) screen[i] = (*sw)[i];
        else warn("'screenwindow' should have four values");
    }
    float fov = camera_p_.find_one_float("fov", 90.0f);
    const float half_fov = camera_p_.find_one_float("halffov", -1.0f);
    if (half_fov > 0.0f) fov = 2.0f * half_fov;
    float r2c[16];
    if (camera_name_ == "environment") {   // environment_camera.rs:86-104: shutter times only
        if (!check(ABI(pbrt_hip_set_camera_environment(scene_, camera_to_world_.m, xres, yres, shutter_open, shutter_close)), "set_camera_environment")) return PBRT_HIP_ERR_INVALID_ARG;
    } else if (camera_name_ == "orthographic") {
        pbrt_hip_host_orthographic_raster_to_camera(xres, yres, screen, r2c);
        if (!check(ABI(pbrt_hip_set_camera_orthographic(scene_, r2c, camera_to_world_.m, lens_radius, focal_distance, shutter_open, shutter_close)), "set_camera_orthographic")) return PBRT_HIP_ERR_INVALID_ARG;
    } else {
        pbrt_hip_host_perspective_raster_to_camera(fov, xres, yres, screen, r2c);
        if (!check(ABI(pbrt_hip_set_camera_perspective(scene_, r2c, camera_to_world_.m, lens_radius, focal_distance, shutter_open, shutter_close)), "set_camera_perspective")) return PBRT_HIP_ERR_INVALID_ARG;
    }

    // ---- sampler (halton.rs:272-290, sobol.rs:201-214)
    int skind;
    if (sampler_name_ == "halton") skind = 0;
    else if (sampler_name_ == "sobol") skind = 1;
    else { error = "Sampler \"" + sampler_name_ + "\" is outside the hot-path scope (supported: halton, sobol)"; return PBRT_HIP_ERR_UNSUPPORTED; }
    const int spp = sampler_p_.find_one_int("pixelsamples", 16);
    if (spp <= 0) { error = "pixelsamples must be positive"; return PBRT_HIP_ERR_INVALID_ARG; }
    if (!check(ABI(pbrt_hip_set_sampler(scene_, skind, (uint32_t)spp, sb, sampler_p_.find_one_bool("samplepixelcenter", false) ? 1 : 0)), "set_sampler")) return PBRT_HIP_ERR_INVALID_ARG;
    if (skind == 1) {
        if (sobol_tables_file.empty()) { error = "Sampler \"sobol\" needs --sobol-tables FILE (the generator matrices are data the library does not embed)"; return PBRT_HIP_ERR_STATE; }
        std::ifstream f(sobol_tables_file, std::ios::binary);
        const size_t n32 = 1024 * 52, nv = 25 * 52, nvi = 26 * 52;
        std::vector<uint32_t> m32(n32); std::vector<uint64_t> vdc(nvi), vdci(nvi);
        if (!f || !f.read((char*)m32.data(), n32 * 4) || !f.read((char*)vdc.data(), nv * 8) || !f.read((char*)vdci.data(), nvi * 8)) {
            error = "cannot read Sobol tables from '" + sobol_tables_file + "'"; return PBRT_HIP_ERR_INVALID_ARG;
        }
        if (!check(ABI(pbrt_hip_set_sobol_tables(scene_, m32.data(), n32, vdc.data(), vdci.data(), nvi)), "set_sobol_tables")) return PBRT_HIP_ERR_INVALID_ARG;
    }

    // ---- accelerator (bvh/mod.rs:339-360)
    int split = 0;
    if (accel_name_ == "bvh") {
        const std::string sm = accel_p_.find_one_string("splitmethod", "sah");
        if (sm == "sah") split = 0;
        else if (sm == "equal") split = 3;
        else if (sm == "hlbvh") split = 1;
        else if (sm == "middle") { error = "BVH splitmethod \"middle\" panics in the reference (sah.rs:67-76) and is not offered (supported: sah, hlbvh, equal)"; return PBRT_HIP_ERR_UNSUPPORTED; }
        else { warn("BVH split method \"" + sm + "\" unknown.  Using \"sah\"."); split = 0; }
    } else { error = "Accelerator \"" + accel_name_ + "\" is outside the hot-path scope (supported: bvh)"; return PBRT_HIP_ERR_UNSUPPORTED; }
    const int max_prims = accel_p_.find_one_int("maxnodeprims", 4);
    if (n_lights_ == 0) warn("No light sources defined in scene; rendering a black image.");
    auto t0 = clk::now();
    // "sah" and "hlbvh" trees of scenes without object instances are made on the GPU (same trees: csrc/bvh_sah_device.hip, bvh_device.hip); everything else by the library's host builder
    int brc = PBRT_HIP_ERR_UNSUPPORTED;
    if ((split == 0 || split == 1) && !check_only_) brc = pbrt_hip_build_accel_device(scene_, split, max_prims);   // UNSUPPORTED for scenes with object definitions: the host builder makes the same tree
    if (brc == PBRT_HIP_ERR_UNSUPPORTED) brc = ABI(pbrt_hip_build_accel(scene_, split, max_prims));
    if (!check(brc, "build_accel")) return PBRT_HIP_ERR_DEVICE;
    rep.build_seconds = std::chrono::duration<double>(clk::now() - t0).count();

    // ---- integrator (path.rs:287-327)
    if (integrator_name_ != "path") { error = "Integrator \"" + integrator_name_ + "\" is outside the hot-path scope (supported: path)"; return PBRT_HIP_ERR_UNSUPPORTED; }
    const int max_depth = integrator_p_.find_one_int("maxdepth", 5);
    int pb[4] = {sb[0], sb[1], sb[2], sb[3]};
    if (const std::vector<int>* v = integrator_p_.find_ints("pixelbounds")) {
        if (v->size() != 4) warn("Expected 4 values for 'pixelbounds' parameter.");
        else {
            pb[0] = std::max(pb[0], (*v)[0]); pb[1] = std::max(pb[1], (*v)[1]);
            pb[2] = std::min(pb[2], (*v)[2]); pb[3] = std::min(pb[3], (*v)[3]);
            if (pb[2] <= pb[0] || pb[3] <= pb[1]) { warn("Degenerate 'pixelbounds' specified."); pb[2] = pb[0]; pb[3] = pb[1]; }
        }
    }
    const float rr = integrator_p_.find_one_float("rrthreshold", 1.0f);
    const std::string lss = integrator_p_.find_one_string("lightsamplestrategy", "spatial");
    int strategy = 2;
    if (lss == "uniform") strategy = 0; else if (lss == "power") strategy = 1; else if (lss == "spatial") strategy = 2;
    else { warn("Light sample distribution type \"" + lss + "\" unknown. Using \"spatial\"."); strategy = 2; }

    rep.xres = xres; rep.yres = yres; std::memcpy(rep.crop, cb, sizeof cb);
    rep.n_triangles = n_tris_; rep.n_lights = n_lights_; rep.n_instances = n_instances_; rep.warnings = warnings;
    rep.spp = spp; rep.max_depth = max_depth; rep.light_strategy = strategy; std::memcpy(rep.pixel_bounds, pb, sizeof pb);
    if (check_only_) { rep.out_file = filename; return PBRT_HIP_OK; }
    const size_t npix = (size_t)std::max(0, cb[2] - cb[0]) * (size_t)std::max(0, cb[3] - cb[1]);
    std::vector<float> xyz(npix * 3), wt(npix), rgb(npix * 3);
    int rc = pbrt_hip_render_path(scene_, max_depth, rr, strategy, pb, tile_size, 0, 1, xyz.data(), wt.data(), &rep.stats);
    if (!check(rc, "render_path")) return rc;
    if (!check(pbrt_hip_film_to_rgb(scene_, xyz.data(), wt.data(), rgb.data()), "film_to_rgb")) return PBRT_HIP_ERR_DEVICE;
    std::string err;
    if (!write_image(filename, rgb.data(), cb[2] - cb[0], cb[3] - cb[1], err)) { error = err; return PBRT_HIP_ERR_INVALID_ARG; }
    rep.out_file = filename; rep.warnings = warnings;
    return PBRT_HIP_OK;
}

// core/src/image_io.rs:336-374: "PF", width height, scale -1 (little endian), scanlines bottom-to-top, f32 RGB.
bool write_pfm(const std::string& path, const float* rgb, int w, int h, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "Unable to open output PFM file '" + path + "'"; return false; }
    std::fprintf(f, "PF\n%d %d\n-1\n", w, h);
    for (int y = h - 1; y >= 0; y--)
        if (std::fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f) != (size_t)w * 3) { err = "Error writing PFM file '" + path + "'"; std::fclose(f); return false; }
    std::fclose(f);
    return true;
}

}  // namespace pbrt_host
