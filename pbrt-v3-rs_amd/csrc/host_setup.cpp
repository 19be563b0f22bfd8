// Host-side helpers that restate the reference's scene set-up arithmetic (the callers of the hot path), exported
// with C linkage so any host (the C++ driver, a Rust shim, Python/ctypes) produces bit-identical camera/film inputs.
//   Transform::{translate,scale,rotate_axis,look_at,perspective}  core/src/geometry/transform.rs:49-210
//   ProjectiveCameraData::new                                      core/src/camera.rs:276-306
//   PerspectiveCamera From<&ParamSet> screen window                cameras/src/perspective_camera.rs:365-407
//   Film::new / get_sample_bounds                                  core/src/film/mod.rs:89-159
#include "../../include/pbrt_hip_host.h"
#include "host_math.h"
#include "guard.h"
#include <algorithm>
using phost::ph_guard;
using phost::ph_guard_void;

using namespace hm;

namespace {
struct Xf { M4 m, mi; };
Xf xf_mul(const Xf& a, const Xf& b) { return {m4_mul(a.m, b.m), m4_mul(b.mi, a.mi)}; }  // transform.rs:644-656
M4 rows(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13, float a20, float a21, float a22,
        float a23, float a30, float a31, float a32, float a33) {
    M4 r; float v[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
    std::memcpy(r.m, v, 64); return r;
}
Xf xf_translate(float x, float y, float z) { return {rows(1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1), rows(1, 0, 0, -x, 0, 1, 0, -y, 0, 0, 1, -z, 0, 0, 0, 1)}; }
Xf xf_scale(float x, float y, float z) {
    return {rows(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1), rows(1.0f / x, 0, 0, 0, 0, 1.0f / y, 0, 0, 0, 0, 1.0f / z, 0, 0, 0, 0, 1)};
}
Xf xf_rotate(float theta, V3 axis) {  // transform.rs:135-163
    V3 a = normalize(axis);
    float r = to_radians(theta), s = std::sin(r), c = std::cos(r);
    M4 m = m4_identity();
    m.m[0][0] = a.x * a.x + (1.0f - a.x * a.x) * c;
    m.m[0][1] = a.x * a.y * (1.0f - c) - a.z * s;
    m.m[0][2] = a.x * a.z * (1.0f - c) + a.y * s;
    m.m[0][3] = 0.0f;
    m.m[1][0] = a.x * a.y * (1.0f - c) + a.z * s;
    m.m[1][1] = a.y * a.y + (1.0f - a.y * a.y) * c;
    m.m[1][2] = a.y * a.z * (1.0f - c) - a.x * s;
    m.m[1][3] = 0.0f;
    m.m[2][0] = a.x * a.z * (1.0f - c) - a.y * s;
    m.m[2][1] = a.y * a.z * (1.0f - c) + a.x * s;
    m.m[2][2] = a.z * a.z + (1.0f - a.z * a.z) * c;
    m.m[2][3] = 0.0f;
    return {m, m4_transpose(m)};
}
void out16(const M4& m, float* o) { if (o) std::memcpy(o, m.m, 64); }
M4 in16(const float* p) { M4 m; std::memcpy(m.m, p, 64); return m; }
}  // namespace

extern "C" {

void pbrt_hip_host_translate(const float d[3], float out_m[16], float out_minv[16]) { Xf t = xf_translate(d[0], d[1], d[2]); out16(t.m, out_m); out16(t.mi, out_minv); }
void pbrt_hip_host_scale(const float s[3], float out_m[16], float out_minv[16]) { Xf t = xf_scale(s[0], s[1], s[2]); out16(t.m, out_m); out16(t.mi, out_minv); }
void pbrt_hip_host_rotate(float theta_deg, const float axis[3], float out_m[16], float out_minv[16]) {
    Xf t = xf_rotate(theta_deg, {axis[0], axis[1], axis[2]}); out16(t.m, out_m); out16(t.mi, out_minv);
}
// Transform * Transform (transform.rs:644-656): out = a*b, out_inv = b_inv*a_inv
void pbrt_hip_host_compose(const float a_m[16], const float a_minv[16], const float b_m[16], const float b_minv[16], float out_m[16], float out_minv[16]) {
    Xf r = xf_mul({in16(a_m), in16(a_minv)}, {in16(b_m), in16(b_minv)}); out16(r.m, out_m); out16(r.mi, out_minv);
}
void pbrt_hip_host_invert(const float m[16], float out[16]) { out16(m4_inverse(in16(m)), out); }

// Transform::look_at (transform.rs:165-189): out_m = world->camera, out_minv = camera->world.  Returns -1 when `up`
// and the viewing direction are parallel (the reference panics).
int pbrt_hip_host_look_at(const float pos[3], const float look[3], const float up[3], float out_m[16], float out_minv[16]) {
    return ph_guard(nullptr, "pbrt_hip_host_look_at", [&]() -> int {
    V3 p = ld(pos), dir = normalize(sub(ld(look), p));
    V3 right = cross(normalize(ld(up)), dir);
    if (len(right) == 0.0f) return -1;
    right = normalize(right);
    V3 new_up = cross(dir, right);
    M4 c2w = rows(right.x, new_up.x, dir.x, p.x, right.y, new_up.y, dir.y, p.y, right.z, new_up.z, dir.z, p.z, 0, 0, 0, 1);
    out16(m4_inverse(c2w), out_m); out16(c2w, out_minv);
    return 0;
    });
}

// Default screen window from the frame aspect ratio (perspective_camera.rs:381-389): {xmin, xmax, ymin, ymax}
void pbrt_hip_host_screen_window(int xres, int yres, float out_screen[4]) {
    float frame = (float)xres / (float)yres;
    if (frame > 1.0f) { out_screen[0] = -frame; out_screen[1] = frame; out_screen[2] = -1.0f; out_screen[3] = 1.0f; }
    else { out_screen[0] = -1.0f; out_screen[1] = 1.0f; out_screen[2] = -1.0f / frame; out_screen[3] = 1.0f / frame; }
}

// raster_to_camera of a PerspectiveCamera: Transform::perspective(fov, 1e-2, 1000) (perspective_camera.rs:60-66) and
// ProjectiveCameraData::new (camera.rs:276-306)
void pbrt_hip_host_perspective_raster_to_camera(float fov_deg, int xres, int yres, const float screen[4], float out_m[16]) {
    const float n = 1e-2f, f = 1000.0f;
    M4 persp = rows(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    float inv_tan_ang = 1.0f / std::tan(to_radians(fov_deg) / 2.0f);
    Xf c2s = xf_mul(xf_scale(inv_tan_ang, inv_tan_ang, 1.0f), Xf{persp, m4_inverse(persp)});
    Xf s2r = xf_mul(xf_mul(xf_scale((float)xres, (float)yres, 1.0f), xf_scale(1.0f / (screen[1] - screen[0]), 1.0f / (screen[2] - screen[3]), 1.0f)),
                    xf_translate(-screen[0], -screen[3], 0.0f));
    Xf r2s{s2r.mi, s2r.m};
    Xf c2s_inv{c2s.mi, c2s.m};
    out16(xf_mul(c2s_inv, r2s).m, out_m);
}

// raster_to_camera of an OrthographicCamera: Transform::orthographic(0, 1) = scale(1, 1, 1 / (far - near)) * translate(0, 0, -near)
// (orthographic_camera.rs:50-56, transform.rs:222-225) through the same ProjectiveCameraData::new
void pbrt_hip_host_orthographic_raster_to_camera(int xres, int yres, const float screen[4], float out_m[16]) {
    const float z_near = 0.0f, z_far = 1.0f;
    Xf c2s = xf_mul(xf_scale(1.0f, 1.0f, 1.0f / (z_far - z_near)), xf_translate(0.0f, 0.0f, -z_near));
    Xf s2r = xf_mul(xf_mul(xf_scale((float)xres, (float)yres, 1.0f), xf_scale(1.0f / (screen[1] - screen[0]), 1.0f / (screen[2] - screen[3]), 1.0f)),
                    xf_translate(-screen[0], -screen[3], 0.0f));
    Xf r2s{s2r.mi, s2r.m};
    Xf c2s_inv{c2s.mi, c2s.m};
    out16(xf_mul(c2s_inv, r2s).m, out_m);
}

// ProjectionLight::new (lights/src/projection.rs:95-111): screen bounds from the image's aspect ratio, light_projection = Transform::perspective(fov, 1e-3, 1e30)
// and the cosine of the frustum's corner direction.  Internal to the library (scene_host.h), used by pbrt_hip_add_light_projection.
extern "C++" {
namespace phost {
void projection_light_setup(float fov_deg, float aspect, float out_proj[16], float out_screen[4], float* out_cos_total_width) {
    if (aspect > 1.0f) { out_screen[0] = -aspect; out_screen[1] = aspect; out_screen[2] = -1.0f; out_screen[3] = 1.0f; }
    else { out_screen[0] = -1.0f; out_screen[1] = 1.0f; out_screen[2] = -1.0f / aspect; out_screen[3] = 1.0f / aspect; }
    const float n = 1e-3f, f = 1e30f;
    M4 persp = rows(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    float inv_tan_ang = 1.0f / std::tan(to_radians(fov_deg) / 2.0f);
    Xf proj = xf_mul(xf_scale(inv_tan_ang, inv_tan_ang, 1.0f), Xf{persp, m4_inverse(persp)});
    out16(proj.m, out_proj);
    const float* m = &proj.mi.m[0][0];
    const float x = out_screen[1], y = out_screen[3], z = 0.0f;   // p_corner; Transform::transform_point (transform.rs:288-302)
    const float xp = m[0] * x + m[1] * y + m[2] * z + m[3], yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    const float zp = m[8] * x + m[9] * y + m[10] * z + m[11], wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    V3 c{xp, yp, zp};
    if (wp != 1.0f) { const float inv = 1.0f / wp; c = V3{inv * xp, inv * yp, inv * zp}; }
    *out_cos_total_width = normalize(c).z;
}
}  // namespace phost
}  // extern "C++"

// Film::new crop bounds (film/mod.rs:101-111), the 16x16 filter table (:113-129) for a BOX filter
// (filters/src/boxf.rs:31-47: evaluate == 1) and Film::get_sample_bounds (:150-159).
void pbrt_hip_host_film_box(int xres, int yres, const float crop_window[4] /*x0 x1 y0 y1*/, const float radius[2], int out_cropped_bounds[4],
                            float out_table[256], int out_sample_bounds[4]) {
    auto sat = [](float v) { if (v != v) return 0; if (v >= 2147483648.0f) return 2147483647; if (v <= -2147483648.0f) return (int)0x80000000; return (int)v; };
    out_cropped_bounds[0] = sat(std::ceil((float)xres * crop_window[0])); out_cropped_bounds[1] = sat(std::ceil((float)yres * crop_window[2]));
    out_cropped_bounds[2] = sat(std::ceil((float)xres * crop_window[1])); out_cropped_bounds[3] = sat(std::ceil((float)yres * crop_window[3]));
    for (int i = 0; i < 256; i++) out_table[i] = 1.0f;
    out_sample_bounds[0] = sat(std::floor((float)out_cropped_bounds[0] + 0.5f - radius[0]));
    out_sample_bounds[1] = sat(std::floor((float)out_cropped_bounds[1] + 0.5f - radius[1]));
    out_sample_bounds[2] = sat(std::ceil((float)out_cropped_bounds[2] - 0.5f + radius[0]));
    out_sample_bounds[3] = sat(std::ceil((float)out_cropped_bounds[3] - 0.5f + radius[1]));
}

// Film::new's 16x16 table (film/mod.rs:113-129) for every reconstruction filter the reference ships:
//   kind 0 box (filters/src/boxf.rs:31-47)            params unused
//   kind 1 gaussian (gaussian.rs:33-62)               params[0] = alpha
//   kind 2 mitchell (mitchell.rs:36-80)               params = {B, C}; keeps the reference's constant term
//                                                     (8*C + 24*C) of the |x|>1 branch as written there (:47)
//   kind 3 sinc / Lanczos (sinc.rs:31-82)             params[0] = tau
//   kind 4 triangle (triangle.rs:28-45)
// Returns -1 for an unknown kind.  exp/sin are the host libm's f32 routines, as they are for the reference's host.
int pbrt_hip_host_film_filter(int kind, const float params[2], int xres, int yres, const float crop_window[4], const float radius[2],
                              int out_cropped_bounds[4], float out_table[256], int out_sample_bounds[4]) {
    return ph_guard(nullptr, "pbrt_hip_host_film_filter", [&]() -> int {
    if (kind < 0 || kind > 4) return -1;
    pbrt_hip_host_film_box(xres, yres, crop_window, radius, out_cropped_bounds, out_table, out_sample_bounds);
    const float rx = radius[0], ry = radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;  // FilterData::new (core/src/filter.rs)
    const float a = params ? params[0] : 0.0f, b = params ? params[1] : 0.0f;
    auto fabs_ = [](float v) { return v < 0.0f ? -v : v; };
    auto fmax_ = [](float x, float y) { return x > y ? x : y; };
    auto gauss = [&](float d, float expv) { return fmax_(0.0f, std::exp(-a * d * d) - expv); };
    auto mitchell = [&](float x0) {
        const float B = a, Cc = b;
        float x = fabs_(2.0f * x0);
        if (x > 1.0f)
            return ((-B - 6.0f * Cc) * x * x * x + (6.0f * B + 30.0f * Cc) * x * x + (-12.0f * B - 48.0f * Cc) * x + (8.0f * Cc + 24.0f * Cc)) * (1.0f / 6.0f);
        return ((12.0f - 9.0f * B - 6.0f * Cc) * x * x * x + (-18.0f + 12.0f * B + 6.0f * Cc) * x * x + (6.0f - 2.0f * B)) * (1.0f / 6.0f);
    };
    const float PI_F = 3.14159265358979323846f;
    auto sinc = [&](float x0) { float x = fabs_(x0); return x < 1e-5f ? 1.0f : std::sin(PI_F * x) / (PI_F * x); };
    auto wsinc = [&](float x0, float r) { float x = fabs_(x0); if (x > r) return 0.0f; float l = sinc(x / a); return sinc(x) * l; };
    const float exp_x = kind == 1 ? std::exp(-a * rx * rx) : 0.0f, exp_y = kind == 1 ? std::exp(-a * ry * ry) : 0.0f;
    const float inv_w = 1.0f / 16.0f;  // INV_FILTER_TABLE_WIDTH
    int o = 0;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++, o++) {
            const float px = ((float)x + 0.5f) * rx * inv_w, py = ((float)y + 0.5f) * ry * inv_w;
            float v = 1.0f;
            switch (kind) {
                case 1: v = gauss(px, exp_x) * gauss(py, exp_y); break;
                case 2: v = mitchell(px * inv_rx) * mitchell(py * inv_ry); break;
                case 3: v = wsinc(px, rx) * wsinc(py, ry); break;
                case 4: v = fmax_(0.0f, rx - fabs_(px)) * fmax_(0.0f, ry - fabs_(py)); break;
                default: break;
            }
            out_table[o] = v;
        }
    return 0;
    });
}

// transform_point / transform_vector / transform_normal (transform.rs:288-302,373-380,441-448) for mesh vertices:
// TriangleMesh::new moves P, N, S to world space once (triangle.rs:93-99).
void pbrt_hip_host_transform_points(const float m[16], const float* in, float* out, size_t n) {
    M4 a = in16(m);
    for (size_t i = 0; i < n; i++) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        float xp = a.m[0][0] * x + a.m[0][1] * y + a.m[0][2] * z + a.m[0][3];
        float yp = a.m[1][0] * x + a.m[1][1] * y + a.m[1][2] * z + a.m[1][3];
        float zp = a.m[2][0] * x + a.m[2][1] * y + a.m[2][2] * z + a.m[2][3];
        float wp = a.m[3][0] * x + a.m[3][1] * y + a.m[3][2] * z + a.m[3][3];
        if (wp == 1.0f) { out[3 * i] = xp; out[3 * i + 1] = yp; out[3 * i + 2] = zp; }
        else { float inv = 1.0f / wp; out[3 * i] = inv * xp; out[3 * i + 1] = inv * yp; out[3 * i + 2] = inv * zp; }
    }
}
void pbrt_hip_host_transform_vectors(const float m[16], const float* in, float* out, size_t n) {
    M4 a = in16(m);
    for (size_t i = 0; i < n; i++) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i] = a.m[0][0] * x + a.m[0][1] * y + a.m[0][2] * z;
        out[3 * i + 1] = a.m[1][0] * x + a.m[1][1] * y + a.m[1][2] * z;
        out[3 * i + 2] = a.m[2][0] * x + a.m[2][1] * y + a.m[2][2] * z;
    }
}
void pbrt_hip_host_transform_normals(const float m_inv[16], const float* in, float* out, size_t n) {
    M4 a = in16(m_inv);
    for (size_t i = 0; i < n; i++) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        out[3 * i] = a.m[0][0] * x + a.m[1][0] * y + a.m[2][0] * z;
        out[3 * i + 1] = a.m[0][1] * x + a.m[1][1] * y + a.m[2][1] * z;
        out[3 * i + 2] = a.m[0][2] * x + a.m[1][2] * y + a.m[2][2] * z;
    }
}
int pbrt_hip_host_swaps_handedness(const float m[16]) {  // transform.rs:593-599
    M4 a = in16(m);
    float det = a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
                a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
    return det < 0.0f ? 1 : 0;
}
// DistantLight::new: w_light = normalize(light_to_world(from - to)) (lights/src/distant.rs:36-50,147-157)
void pbrt_hip_host_distant_direction(const float l2w[16], const float from[3], const float to[3], float out_w[3]) {
    float d[3] = {from[0] - to[0], from[1] - to[1], from[2] - to[2]}, t[3];
    pbrt_hip_host_transform_vectors(l2w, d, t, 1);
    V3 w = normalize(ld(t));
    out_w[0] = w.x; out_w[1] = w.y; out_w[2] = w.z;
}
// PointLight: p_light = (Translate(from) * light_to_world)(0,0,0) (lights/src/point.rs:36-55,150-158)
void pbrt_hip_host_point_position(const float l2w[16], const float l2w_inv[16], const float from[3], float out_p[3]) {
    Xf t = xf_mul(xf_translate(from[0], from[1], from[2]), Xf{in16(l2w), in16(l2w_inv)});
    float z[3] = {0, 0, 0}, m[16];
    out16(t.m, m);
    pbrt_hip_host_transform_points(m, z, out_p, 1);
}

}  // extern "C"

// ---- synthetic measurement scene (SURVEY §8d "gen_random_tris") ---------------------------------------------
// PCG32 stream `seed` (RNG::new(seed), core/src/rng.rs:39-62).  Per triangle: centre c ~ U[-1,1)^3, then three vertices
// c + s*U[-1,1)^3 with s = 1.5 * N^(-1/3) (constant expected density), U = 2*uniform_float()-1.  Unshared vertices
// (3N positions, indices 0..3N-1).
extern "C" // SpotLight From<ParamSet> (lights/src/spot.rs:150-190): light_to_world = ctm * translate(from) * inverse(dir_to_z) with dir_to_z rows
// (du, dv, dir) from coordinate_system(normalize(to - from)); cos_total_width = cos(radians(coneangle)), cos_falloff_start =
// cos(radians(coneangle - conedeltaangle)) (:42-43, host libm cosf as for the reference).
void pbrt_hip_host_spot(const float ctm_m[16], const float ctm_minv[16], const float from[3], const float to[3], float cone_angle, float cone_delta,
                        float out_l2w[16], float out_w2l[16], float out_cos[2]) {
    V3 dir = normalize(sub({to[0], to[1], to[2]}, {from[0], from[1], from[2]}));
    V3 du, dv;
    coordinate_system(dir, du, dv);
    Xf d2z; d2z.m = rows(du.x, du.y, du.z, 0, dv.x, dv.y, dv.z, 0, dir.x, dir.y, dir.z, 0, 0, 0, 0, 1); d2z.mi = m4_inverse(d2z.m);
    Xf inv_d2z = {d2z.mi, d2z.m};
    Xf r = xf_mul(xf_mul({in16(ctm_m), in16(ctm_minv)}, xf_translate(from[0], from[1], from[2])), inv_d2z);
    out16(r.m, out_l2w); out16(r.mi, out_w2l);
    out_cos[0] = std::cos(to_radians(cone_angle));
    out_cos[1] = std::cos(to_radians(cone_angle - cone_delta));
}

void pbrt_hip_host_gen_random_tris(uint64_t n_tris, uint64_t seed, float* out_P /*9 per tri*/, uint32_t* out_idx /*3 per tri*/) {
    ph_guard_void([&]() {
    Pcg32 rng;
    rng.state = 0; rng.inc = (seed << 1) | 1;  // RNG::set_sequence
    rng.next(); rng.state += 0x853c49e6748fea9bULL; rng.next();
    auto uf = [&]() { float u = (float)rng.next() * 0x1.0p-32f; return u < 0x1.fffffep-1f ? u : 0x1.fffffep-1f; };  // uniform_float (rng.rs:98-103)
    const float s = 1.5f * std::pow((float)n_tris, -1.0f / 3.0f);
    for (uint64_t t = 0; t < n_tris; t++) {
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = 2.0f * uf() - 1.0f;
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) out_P[9 * t + 3 * v + k] = c[k] + s * (2.0f * uf() - 1.0f);
        out_idx[3 * t] = (uint32_t)(3 * t); out_idx[3 * t + 1] = (uint32_t)(3 * t + 1); out_idx[3 * t + 2] = (uint32_t)(3 * t + 2);
    }
    });
}

// ---- host-only view of the BVH builder (no device needed): used by the CPU test-suite to pin the topology contract of
// bvh_build.h against the oracle's restatement of BVHAccel::new.  out_ordered_prims: n_tris entries (leaf order);
// out_nodes: up to n_tris-1 Node64 (may be NULL); out_info: {interior nodes, leaf nodes, max prims in a leaf, max depth, root_ref}.
#include "bvh_build.h"
extern "C" int pbrt_hip_host_build_bvh(const float* P, const uint32_t* idx, uint64_t n_tris, int split_method, int max_prims_in_node, int n_threads,
                                       uint32_t* out_ordered_prims, uint32_t* out_leaf_last, void* out_nodes, uint64_t* out_info, float* out_root_bounds) {
    return ph_guard(nullptr, "pbrt_hip_host_build_bvh", [&]() -> int {
    phost::BuildInput in{P, idx, (size_t)n_tris, nullptr, nullptr};
    phost::BuildOutput out;
    int rc = phost::build_bvh(in, split_method, max_prims_in_node, n_threads, out);
    if (rc) return rc;
    for (size_t i = 0; i < out.tris.size(); i++) {
        if (out_ordered_prims) out_ordered_prims[i] = out.tris[i].prim;
        if (out_leaf_last) out_leaf_last[i] = (out.tris[i].flags & PH_TRI_LAST) ? 1u : 0u;
    }
    if (out_nodes && !out.nodes.empty()) std::memcpy(out_nodes, out.nodes.data(), out.nodes.size() * sizeof(Node64));
    if (out_info) { out_info[0] = out.interior_nodes; out_info[1] = out.leaf_nodes; out_info[2] = out.max_leaf_prims; out_info[3] = (uint64_t)out.max_depth; out_info[4] = out.root_ref; }
    if (out_root_bounds) { for (int k = 0; k < 3; k++) { out_root_bounds[k] = out.root_lo[k]; out_root_bounds[3 + k] = out.root_hi[k]; } }
    return 0;
    });
}

// ---- spectral parameter types of the scene description -> RGB ------------------------------------------------------------------------------------
// ParamSet::add_blackbody_spectrum / add_sampled_spectrum (core/src/paramset/mod.rs:236-262), blackbody / blackbody_normalized / interpolate_spectrum_samples
// (core/src/spectrum/common.rs:315-398), RGBSpectrum::from(&Vec<Sample>) (core/src/spectrum/rgb_spectrum.rs:82-103), all in f32 as the reference computes them.
#include "cie_table.inc"
#include "copper_table.inc"
#include <cmath>
#include <vector>
namespace {
float planck(float lambda_nm, float t) {  // spectrum/common.rs:361-380, one wavelength
    const float c = 299792458.0f, h = 6.62606957e-34f, kb = 1.3806488e-23f;
    const float l = lambda_nm * 1e-9f;
    const float lambda5 = (l * l) * (l * l) * l;
    return (2.0f * h * c * c) / (lambda5 * (std::exp((h * c) / (l * kb * t)) - 1.0f));
}
size_t find_interval_sz(size_t size, const std::vector<std::pair<float, float>>& s, float l) {  // pbrt/common.rs:251-276 with the predicate `samples[i].lambda <= l`
    size_t first = 0, len = size;
    while (len > 0) {
        const size_t half = len >> 1, middle = first + half;
        if (s[middle].first <= l) { first = middle + 1; len -= half + 1; } else len = half;
    }
    if (first == 0) return 0;
    const size_t v = first - 1;
    return v > size - 2 ? size - 2 : v;
}
float interpolate_samples(const std::vector<std::pair<float, float>>& s, float l) {  // spectrum/common.rs:315-331
    const size_t n = s.size();
    if (l <= s[0].first) return s[0].second;
    if (l >= s[n - 1].first) return s[n - 1].second;
    const size_t o = find_interval_sz(n, s, l);
    const float t = (l - s[o].first) / (s[o + 1].first - s[o].first);
    return (1.0f - t) * s[o].second + t * s[o + 1].second;
}
void samples_to_rgb(const std::vector<std::pair<float, float>>& s, float out[3]) {  // rgb_spectrum.rs:82-103 + xyz_to_rgb
    float xyz[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 471; i++) {
        const float val = interpolate_samples(s, (float)(360 + i));
        xyz[0] += val * kCieXyz[i][0]; xyz[1] += val * kCieXyz[i][1]; xyz[2] += val * kCieXyz[i][2];
    }
    const float scale = (float)(830 - 360) / (kCieYIntegral * (float)471);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    out[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    out[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    out[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
}  // namespace

extern "C" void pbrt_hip_host_blackbody_rgb(float temperature, float scale, float out_rgb[3]) {
    std::vector<std::pair<float, float>> s(471);
    float mx = 1.0f;
    if (temperature > 0.0f) mx = planck(2.8977721e-3f / temperature * 1e9f, temperature);  // Wien's displacement law, metres -> nanometres
    for (int i = 0; i < 471; i++) {
        const float lam = (float)(360 + i);
        s[i] = {lam, temperature > 0.0f ? planck(lam, temperature) / mx : 0.0f};
    }
    samples_to_rgb(s, out_rgb);
    for (int k = 0; k < 3; k++) out_rgb[k] = scale * out_rgb[k];
}

// (wavelength nm, value) pairs.  Unsorted input is sorted by wavelength first: the reference sorts a copy and then interpolates the ORIGINAL order, where its
// interval assertion fails (rgb_spectrum.rs:84-91, common.rs:327) — there is no result of the reference to match for unsorted samples.  Returns -1 for n == 0.
extern "C" int pbrt_hip_host_sampled_rgb(const float* lambda_value_pairs, size_t n_samples, float out_rgb[3]) {
    return ph_guard(nullptr, "pbrt_hip_host_sampled_rgb", [&]() -> int {
    if (!lambda_value_pairs || n_samples == 0) return -1;
    std::vector<std::pair<float, float>> s(n_samples);
    for (size_t i = 0; i < n_samples; i++) s[i] = {lambda_value_pairs[2 * i], lambda_value_pairs[2 * i + 1]};
    std::stable_sort(s.begin(), s.end(), [](const std::pair<float, float>& a, const std::pair<float, float>& b) { return a.first < b.first; });
    if (n_samples == 1) { s.push_back(s[0]); }  // a single sample is a constant spectrum (both clamps of interpolate_spectrum_samples return it)
    samples_to_rgb(s, out_rgb);
    return 0;
    });
}

// MetalMaterial's defaults (materials/src/metal.rs:136-147): RGB of the copper n and k spectra
extern "C" void pbrt_hip_host_copper_rgb(float out_eta[3], float out_k[3]) {
    std::vector<std::pair<float, float>> n(56), k(56);
    for (int i = 0; i < 56; i++) { n[i] = {kCopper[i][0], kCopper[i][1]}; k[i] = {kCopper[i][0], kCopper[i][2]}; }
    samples_to_rgb(n, out_eta); samples_to_rgb(k, out_k);
}
