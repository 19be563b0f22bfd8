/*
 * pbrt_hip_host.h — optional host-side helpers exported by libpbrt_hip.so.
 *
 * They restate, in the reference's exact f32 expression order, the scene set-up arithmetic that sits on the
 * CALLER side of the hot path (api/src/lib.rs CTM handling, camera/film construction), so that a host written in
 * any language feeds pbrt_hip_set_camera_perspective / pbrt_hip_set_film / pbrt_hip_add_mesh the same bits the
 * reference would compute.  A Rust host does not need them (it already owns these values); the C++ driver and
 * the Python test/bench harness do.  Matrices are row-major float[16].
 */
#ifndef PBRT_HIP_HOST_H
#define PBRT_HIP_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Transform::translate / scale / rotate_axis (core/src/geometry/transform.rs:49-85,135-163) */
void pbrt_hip_host_translate(const float d[3], float out_m[16], float out_minv[16]);
void pbrt_hip_host_scale(const float s[3], float out_m[16], float out_minv[16]);
void pbrt_hip_host_rotate(float theta_deg, const float axis[3], float out_m[16], float out_minv[16]);
/* Transform * Transform (transform.rs:644-656) */
void pbrt_hip_host_compose(const float a_m[16], const float a_minv[16], const float b_m[16], const float b_minv[16],
                           float out_m[16], float out_minv[16]);
/* Matrix4x4::inverse (core/src/geometry/matrix4x4.rs:67-143) */
void pbrt_hip_host_invert(const float m[16], float out[16]);
/* Transform::look_at (transform.rs:165-189): out_m world->camera, out_minv camera->world; -1 if degenerate */
int pbrt_hip_host_look_at(const float pos[3], const float look[3], const float up[3], float out_m[16], float out_minv[16]);
/* default screen window (cameras/src/perspective_camera.rs:381-389): {xmin, xmax, ymin, ymax} */
void pbrt_hip_host_screen_window(int xres, int yres, float out_screen[4]);
/* PerspectiveCamera::new + ProjectiveCameraData::new (perspective_camera.rs:47-66, core/src/camera.rs:276-306) */
void pbrt_hip_host_perspective_raster_to_camera(float fov_deg, int xres, int yres, const float screen[4], float out_m[16]);
/* OrthographicCamera::new + ProjectiveCameraData::new (orthographic_camera.rs:39-56, transform.rs:222-225, core/src/camera.rs:276-306) */
void pbrt_hip_host_orthographic_raster_to_camera(int xres, int yres, const float screen[4], float out_m[16]);
/* Film::new with a BoxFilter + Film::get_sample_bounds (core/src/film/mod.rs:89-159, filters/src/boxf.rs) */
void pbrt_hip_host_film_box(int xres, int yres, const float crop_window[4], const float radius[2], int out_cropped_bounds[4],
                            float out_table[256], int out_sample_bounds[4]);
/* Same for every filter of filters/src: kind 0 box, 1 gaussian {alpha}, 2 mitchell {B,C}, 3 sinc {tau}, 4 triangle
 * (boxf.rs:31-47, gaussian.rs:33-62, mitchell.rs:36-80, sinc.rs:31-82, triangle.rs:28-45).  -1 on an unknown kind. */
int pbrt_hip_host_film_filter(int kind, const float params[2], int xres, int yres, const float crop_window[4],
                              const float radius[2], int out_cropped_bounds[4], float out_table[256], int out_sample_bounds[4]);
/* Transform::transform_point / _vector / _normal (transform.rs:288-302,373-380,441-448), n elements of 3 floats */
void pbrt_hip_host_transform_points(const float m[16], const float* in, float* out, size_t n);
void pbrt_hip_host_transform_vectors(const float m[16], const float* in, float* out, size_t n);
void pbrt_hip_host_transform_normals(const float m_inv[16], const float* in, float* out, size_t n);
int pbrt_hip_host_swaps_handedness(const float m[16]); /* transform.rs:593-599 */
/* DistantLight::new / PointLight From<ParamSet> (lights/src/distant.rs:36-50,147-157; point.rs:36-55,150-158) */
void pbrt_hip_host_distant_direction(const float l2w[16], const float from[3], const float to[3], float out_w[3]);
void pbrt_hip_host_point_position(const float l2w[16], const float l2w_inv[16], const float from[3], float out_p[3]);

/* SpotLight From<ParamSet> (lights/src/spot.rs:150-190): light_to_world / world_to_light from the CTM, `from`, `to`; out_cos =
 * {cos_total_width, cos_falloff_start} from coneangle / conedeltaangle (degrees). */
void pbrt_hip_host_spot(const float ctm_m[16], const float ctm_minv[16], const float from[3], const float to[3], float cone_angle,
                        float cone_delta, float out_l2w[16], float out_w2l[16], float out_cos[2]);

/* Spectral parameter types of a scene description -> RGB, as the reference's ParamSet turns them into RGBSpectrum (core/src/paramset/mod.rs:236-262,
 * core/src/spectrum/common.rs:315-398, rgb_spectrum.rs:82-103): `"blackbody L" [T scale]` and `"spectrum name" [lambda value ...]` (or the pairs of an SPD file). */
void pbrt_hip_host_blackbody_rgb(float temperature_kelvin, float scale, float out_rgb[3]);
int pbrt_hip_host_sampled_rgb(const float* lambda_value_pairs, size_t n_samples, float out_rgb[3]);
/* `Material "metal"` without `eta` / `k`: the reference's copper spectra as RGB (materials/src/metal.rs:136-147) */
void pbrt_hip_host_copper_rgb(float out_eta[3], float out_k[3]);

/* Synthetic measurement scene of BASELINE.md §3: n_tris random triangles from PCG32 stream `seed`
 * (core/src/rng.rs semantics). out_P: 9 floats per triangle, out_idx: 3 per triangle (unshared vertices). */
void pbrt_hip_host_gen_random_tris(uint64_t n_tris, uint64_t seed, float* out_P, uint32_t* out_idx);

/* Host-only run of the library's BVH builder (accelerators/src/bvh/{mod,sah}.rs topology contract), for inspection
 * and tests; needs no GPU.  out_ordered_prims / out_leaf_last: n_tris entries in leaf order; out_nodes: room for
 * n_tris-1 64-byte nodes or NULL; out_info[5] = {interior nodes, leaf nodes, max prims per leaf, max depth, root ref};
 * out_root_bounds[6]. */
int pbrt_hip_host_build_bvh(const float* P, const uint32_t* idx, uint64_t n_tris, int split_method, int max_prims_in_node,
                            int n_threads, uint32_t* out_ordered_prims, uint32_t* out_leaf_last, void* out_nodes,
                            uint64_t* out_info, float* out_root_bounds);
/* The same outputs from the device builders (csrc/bvh_sah_device.hip: split_method 0, csrc/bvh_device.hip: split_method 1) on GPU `device`.  out_seconds: wall time of the build incl. transfers. */
int pbrt_hip_device_build_bvh(int device, const float* P, const uint32_t* idx, uint64_t n_tris, int split_method, int max_prims_in_node, uint32_t* out_ordered_prims,
                              uint32_t* out_leaf_last, void* out_nodes, uint64_t* out_info, float* out_root_bounds, double* out_seconds);

#ifdef __cplusplus
}
#endif
#endif
