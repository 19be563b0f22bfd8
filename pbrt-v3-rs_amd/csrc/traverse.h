// K2/K3/K5 — BVH traversal + watertight triangle test for gfx950 (wave64).
//
// Replaces BVHAccel::intersect / intersect_p (accelerators/src/bvh/mod.rs:173-283), Bounds3::intersect_p_inv
// (core/src/geometry/bounds3.rs:292-325) and the accept/reject part of Triangle::intersect / intersect_p
// (shapes/src/triangle.rs:438-545, 731-837), with GeometricPrimitive::intersect's `r.t_max = t`
// (core/src/primitives/geometric_primitive.rs:67-88).
//
// Equivalence argument (why results are bit-identical to the reference, not merely "as good"):
//  * same tree topology and leaf order (host builder restates sah.rs), same child visit order (dir_is_neg[axis]);
//  * a child's box is tested when its PARENT is visited instead of when the child is popped.  The box test depends on
//    the ray's t_max only through its last clause `t_min < ray.t_max`; the entry distance t_min is kept on the stack and
//    that clause is re-evaluated with the current t_max when the entry is popped -> same visit/skip decision;
//  * box and triangle arithmetic is the reference's expression order, compiled with -ffp-contract=off.  A node step's twelve subtractions and multiplications are issued as six +
//    six packed two-float operations on (lo, hi) pairs and the near / far choice is made on the products (node_boxes): the same operands meet in the same IEEE operations;
//  * stack entries are popped in the reference's order and tested against the t_max of the moment the reference would test them — one entry per node step (deferred pops: a lane
//    that owes a pop is at no leaf, so its t_max cannot change in between);
//  * the flat kernels remember an accepted hit by its TriRec alone and recompute its barycentrics when the ray retires (tri_bary: the test's own e / det / inv_det operations).
//
// Execution shape: persistent waves pull rays from a global counter (one atomic per refill, wave-aggregated with
// ballot/popcount), per-lane replacement of finished rays, while-while traversal, traversal stack in LDS
// ([depth][lane] so every push/pop is conflict-free) spilling to a per-lane global region beyond PH_LDS_DEPTH.
#pragma once
#include "dmath.h"
#include "scene_types.h"

namespace ph {

struct alignas(16) RayIn { float ox, oy, oz, t_max, dx, dy, dz, time; };           // = PbrtHipRay
struct alignas(16) HitOut { float t; uint32_t prim; float b0, b1, b2; uint32_t pad[3]; };  // = PbrtHipHit

#ifndef PH_TRAV_BLOCK
#define PH_TRAV_BLOCK 256
#endif
#ifndef PH_LDS_DEPTH
#define PH_LDS_DEPTH 12
#endif
#define PH_MAX_STACK 64  // the reference's nodes_to_visit[64] (bvh/mod.rs:185)

// Rays, queue order and results are touched once per launch: they are read / written with the non-temporal hint so that they do not displace tree nodes from the caches
// (configs[2] / configs[3] traversal 705.4 -> 701.9 / 735.8 -> 732.4 ms per frame, same-box A/B gpurun r03al; -DPH_STREAM_NT=0 is the plain form).
#ifndef PH_STREAM_NT
#define PH_STREAM_NT 1
#endif
#if PH_STREAM_NT && defined(__HIP_DEVICE_COMPILE__)
typedef float ph_v4f __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float4 ph_stream_load(const float4* p) { const ph_v4f v = __builtin_nontemporal_load(reinterpret_cast<const ph_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
static __device__ __forceinline__ uint32_t ph_stream_load(const uint32_t* p) { return __builtin_nontemporal_load(p); }
static __device__ __forceinline__ void ph_stream_store(float4 v, float4* p) { const ph_v4f w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<ph_v4f*>(p)); }
#define PH_STREAM_LOAD(p) ph_stream_load(p)
#define PH_STREAM_STORE(v, p) ph_stream_store(v, p)
#else
#define PH_STREAM_LOAD(p) (*(p))
#define PH_STREAM_STORE(v, p) (*(p) = (v))
#endif
struct TravParams {
    const RayIn* rays;
    void* out;              // HitOut* (closest) or uint8_t* (any-hit)
    uint32_t n;             // ray count, or (n_ptr != nullptr) read from device memory: no host sync between wavefront stages
    const uint32_t* n_ptr;
    uint32_t* counter;      // work-queue head, zeroed before launch
    uint2* spill;           // [PH_MAX_STACK - PH_LDS_DEPTH][total_threads]
    uint32_t total_threads;
    uint32_t* error_flag;   // set to 1 on stack overflow (the reference would panic on index 64)
    uint32_t batch;         // rays a wave claims per global atomic
    unsigned long long* counts;  // COUNT builds only: [0] interior nodes whose box test passed, [1] triangle tests, [2] rays
    // MIXED kernels only: one launch walks the closest-hit queue (rays/out/n_ptr) and then the any-hit queue below
    const RayIn* rays2; uint8_t* out2; const uint32_t* n2_ptr;
    // optional (wavefront rounds): `order` = the queue positions regrouped by ray-origin cell (raysort.h), walked instead of 0, 1, 2, ...
    const uint32_t* order;
    // optional: n_heads > 1 queue heads, head h serving the head_chunk-ray chunks c with c % n_heads == h.  A block starts on head
    // blockIdx % n_heads — blocks b and b + 8 share an XCD and its L2 (MI355X_MICROARCH.md) — so with chunks about as long as an XCD's share of the
    // rays in flight (32 CUs x 6 blocks x 256 lanes = 48 k) one L2 serves ONE contiguous stretch of the (binned) queue instead of an eighth of
    // every stretch, and all heads still advance through the queue at the same pace (no per-segment tails).  heads[16 * h]; head_chunk % batch == 0.
    uint32_t* heads; uint32_t n_heads, head_chunk;
    unsigned long long* phase;   // PH_PHASE_CLOCK builds only (measurement): [phase] cycles, [12 + phase] executions, [24 + phase] active lanes, summed over all waves
};

struct RayState {
    float ox, oy, oz, t_max;   // (the direction is not kept: after ray_setup only its reciprocals, signs and the shear are used; the instancing kernel keeps the scene-level one beside it)
    float ix, iy, iz;       // 1/d (three IEEE divides, bvh/mod.rs:176)
    uint32_t sgn;           // bits 0..2: dir_is_neg x, y, z (bvh/mod.rs:177-181); bits 3..4: kz of the triangle test (triangle.rs:457-459: kx = kz + 1, ky = kx + 1, both modulo 3)
    float sx, sy, sz;       // triangle.rs:467-469
    PH_DEV bool nx() const { return (sgn & 1u) != 0u; }
    PH_DEV bool ny() const { return (sgn & 2u) != 0u; }
    PH_DEV bool nz() const { return (sgn & 4u) != 0u; }
    PH_DEV int kz() const { return (int)((sgn >> 3) & 3u); }
    PH_DEV int kx() const { const int k = kz() + 1; return k == 3 ? 0 : k; }
    PH_DEV int ky() const { const int k = kx() + 1; return k == 3 ? 0 : k; }
};

PH_DEV void ray_setup(RayState& r, const RayIn& in) {
    r.ox = in.ox; r.oy = in.oy; r.oz = in.oz; r.t_max = in.t_max;
    r.ix = 1.0f / in.dx; r.iy = 1.0f / in.dy; r.iz = 1.0f / in.dz;
    f3 d = mk3(in.dx, in.dy, in.dz);
    const int kz = max_dimension(vabs(d));
    r.sgn = (r.ix < 0.0f ? 1u : 0u) | (r.iy < 0.0f ? 2u : 0u) | (r.iz < 0.0f ? 4u : 0u) | ((uint32_t)kz << 3);
    f3 dp = permute(d, r.kx(), r.ky(), kz);
    r.sx = -dp.x / dp.z; r.sy = -dp.y / dp.z; r.sz = 1.0f / dp.z;
}

// Transform::transform_ray (core/src/geometry/transform.rs:451-476) by a row-major 4x4 `c`: transform_point_with_error on the
// origin (:304-328), transform_vector on the direction, origin pushed to the edge of its error box and t_max shortened by the
// same dt (quirk B2).  Used when a ray enters an object instance (transformed_primitive.rs:51-53).
PH_DEV RayIn xf_ray(const float* c, const RayState& r, f3 rd, float time) {
    const float x = r.ox, y = r.oy, z = r.oz;
    const float ox = (c[0] * x + c[1] * y) + (c[2] * z + c[3]);
    const float oy = (c[4] * x + c[5] * y) + (c[6] * z + c[7]);
    const float oz = (c[8] * x + c[9] * y) + (c[10] * z + c[11]);
    const float ow = (c[12] * x + c[13] * y) + (c[14] * z + c[15]);
    const float xs = pabs(c[0] * x) + pabs(c[1] * y) + pabs(c[2] * z) + pabs(c[3]);
    const float ys = pabs(c[4] * x) + pabs(c[5] * y) + pabs(c[6] * z) + pabs(c[7]);
    const float zs = pabs(c[8] * x) + pabs(c[9] * y) + pabs(c[10] * z) + pabs(c[11]);
    const f3 o_err = kGamma3 * mk3(xs, ys, zs);
    f3 o = (ow == 1.0f) ? mk3(ox, oy, oz) : mk3(ox, oy, oz) / ow;
    const f3 d = mk3(c[0] * rd.x + c[1] * rd.y + c[2] * rd.z, c[4] * rd.x + c[5] * rd.y + c[6] * rd.z, c[8] * rd.x + c[9] * rd.y + c[10] * rd.z);
    const float l2 = length_squared(d);
    float t_max = r.t_max;
    if (l2 > 0.0f) {
        const float dt = ph_div(dot(vabs(d), o_err), l2);
        o = o + d * dt;
        t_max -= dt;
    }
    RayIn out; out.ox = o.x; out.oy = o.y; out.oz = o.z; out.t_max = t_max; out.dx = d.x; out.dy = d.y; out.dz = d.z; out.time = time;
    return out;
}

// v_max3_f32 / v_min3_f32 as the hardware defines them (IEEE mode: a quiet NaN operand is ignored, the result is the extreme of the others)
PH_DEV float vmax3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
#endif
}
PH_DEV float vmin3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return a < b ? (a < c ? a : c) : (b < c ? b : c);
#endif
}

// Bounds3::intersect_p_inv without its final `t_min < ray.t_max` clause; returns t_min via reference.
// Quirk B1 (z far plane not widened) is reproduced.  The reference interleaves four "miss" comparisons with four conditional updates of t_min / t_max
// (bounds3.rs:299-323); here the three entry distances go through one v_max3_f32, the three exit distances through one v_min3_f32 and the verdict is
// `t_min <= t_max`.  Same answer in every case:
//  * numbers: the reference's tests are `entry_i > exit_j -> miss` for the pairs (x,y) (y,x) (xy,z) (z,xy); with the merges in between that is
//    "every entry <= every exit of ANOTHER axis".  max3 <= min3 adds the same-axis pairs: z's always holds (unscaled, monotone arithmetic), x's or y's
//    can only fail when the exit distance is negative (the 1 + 2 gamma_3 factor moves a negative exit below its entry), and then the reference misses
//    too through its final `t_max > 0`.  The merged t_min / t_max are the same numbers (the sign of a zero is not observable: they are only compared).
//  * NaNs (0 x inf: a ray parallel to a slab with its origin on the slab's plane): every comparison with a NaN is false.  A NaN y or z distance
//    therefore neither misses nor replaces t_min / t_max in the reference — exactly what the hardware's max3 / min3 do with a quiet NaN operand.  A NaN
//    x distance is what t_min / t_max START as, survives every update and fails the final comparison: the reference misses; hence the `ordered` term.
PH_DEV bool box_test(const RayState& r, float xn, float xf, float yn, float yf, float zn, float zf, float& t_min_out) {
    // (the same arithmetic as three + three packed two-float operations, v_pk_add_f32 / v_pk_mul_f32, was measured in round 2: more registers, no gain — HISTORY §7b)
    float t_x_min = (xn - r.ox) * r.ix;
    float t_x_max = (xf - r.ox) * r.ix;
    const float t_y_min = (yn - r.oy) * r.iy;
    float t_y_max = (yf - r.oy) * r.iy;
    t_x_max *= kBoxScale;
    t_y_max *= kBoxScale;
    const float t_z_min = (zn - r.oz) * r.iz;
    const float t_z_max = (zf - r.oz) * r.iz;
    const float t_min = vmax3(t_x_min, t_y_min, t_z_min);
    const float t_max = vmin3(t_x_max, t_y_max, t_z_max);
    t_min_out = t_min;
    return (t_min <= t_max) & (t_max > 0.0f) & ((t_x_min == t_x_min) & (t_x_max == t_x_max));
}

// Both boxes of a Node64 against one ray: box_test twice (incl. its caller's `t_min < ray.t_max`), written for the packed two-float VALU operations of gfx950.
// A node stores each slab as the pair (lo, hi) in adjacent dwords, so (lo - o) * inv and (hi - o) * inv are ONE v_pk_add_f32 and ONE v_pk_mul_f32 per axis and box, on
// the registers the load filled; the near / far choice (dir_is_neg, bounds3.rs:296-298 indexes the bounds with it) is made on the two products afterwards — the same
// operands meet in the same operations, so every product has the bits box_test's have —, the widening factor goes onto (far x, far y) as one more packed multiply.
// 15 VALU instructions per box instead of 22.  The NaN test of the x slab is symmetric in (near, far) and widening keeps a NaN a NaN and a number a number, so it reads the raw pair.
#ifndef PH_NODE_PK
#define PH_NODE_PK 1
#endif
typedef float ph_v2f __attribute__((ext_vector_type(2)));
// A per-lane condition as the wave's lane mask (the active lanes' bits) and back: conditions combined as masks cost scalar instructions, not vector ones.
typedef unsigned long long ph_mask;
PH_DEV ph_mask lanes(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(c);
#else
    return c ? 1ull : 0ull;
#endif
}
PH_DEV bool lane_of(ph_mask m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_inverse_ballot_w64(m);
#else
    return (m & 1ull) != 0ull;
#endif
}
// (lo, hi) - o and * inv with the ray's scalar taken from one half of a register pair for BOTH results (op_sel), so the ray needs no duplicated registers:
// SEL = 0 reads the pair's low dword, 1 its high dword.  a - b is a + (-b) in IEEE arithmetic (neg_lo / neg_hi).
template <int SEL> PH_DEV ph_v2f pk_sub_bcast(ph_v2f a, ph_v2f p) {
#if defined(__HIP_DEVICE_COMPILE__)
    ph_v2f d;
    if (SEL == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(p));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(p));
    return d;
#else
    const float o = SEL ? p.y : p.x; return (ph_v2f){a.x - o, a.y - o};
#endif
}
template <int SEL> PH_DEV ph_v2f pk_mul_bcast(ph_v2f a, ph_v2f p) {
#if defined(__HIP_DEVICE_COMPILE__)
    ph_v2f d;
    if (SEL == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(p));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(p));
    return d;
#else
    const float o = SEL ? p.y : p.x; return (ph_v2f){a.x * o, a.y * o};
#endif
}
PH_DEV void node_boxes(const RayState& r, float4 q0, float4 q1, float4 q2, ph_mask& m0, float& t0, ph_mask& m1, float& t1) {
    const ph_v2f pxy = {r.ox, r.oy}, pzi = {r.oz, r.ix}, pii = {r.iy, r.iz};
    const ph_v2f ax = pk_mul_bcast<1>(pk_sub_bcast<0>((ph_v2f){q0.x, q0.y}, pxy), pzi), ay = pk_mul_bcast<0>(pk_sub_bcast<1>((ph_v2f){q0.z, q0.w}, pxy), pii),
                 az = pk_mul_bcast<1>(pk_sub_bcast<0>((ph_v2f){q1.x, q1.y}, pzi), pii);
    const ph_v2f bx = pk_mul_bcast<1>(pk_sub_bcast<0>((ph_v2f){q1.z, q1.w}, pxy), pzi), by = pk_mul_bcast<0>(pk_sub_bcast<1>((ph_v2f){q2.x, q2.y}, pxy), pii),
                 bz = pk_mul_bcast<1>(pk_sub_bcast<0>((ph_v2f){q2.z, q2.w}, pzi), pii);
    const bool nx = r.nx(), ny = r.ny(), nz = r.nz();
    const ph_v2f scale = {kBoxScale, kBoxScale};
    const ph_v2f a_far = (ph_v2f){nx ? ax.x : ax.y, ny ? ay.x : ay.y} * scale;
    const ph_v2f b_far = (ph_v2f){nx ? bx.x : bx.y, ny ? by.x : by.y} * scale;
    t0 = vmax3(nx ? ax.y : ax.x, ny ? ay.y : ay.x, nz ? az.y : az.x);
    const float a_max = vmin3(a_far.x, a_far.y, nz ? az.x : az.y);
    t1 = vmax3(nx ? bx.y : bx.x, ny ? by.y : by.x, nz ? bz.y : bz.x);
    const float b_max = vmin3(b_far.x, b_far.y, nz ? bz.x : bz.y);
    // the verdicts as lane masks (one scalar AND per clause; kept as per-lane booleans they travel through vector registers)
    m0 = lanes(t0 <= a_max) & lanes(a_max > 0.0f) & lanes(!__builtin_isunordered(ax.x, ax.y)) & lanes(t0 < r.t_max);
    m1 = lanes(t1 <= b_max) & lanes(b_max > 0.0f) & lanes(!__builtin_isunordered(bx.x, bx.y)) & lanes(t1 < r.t_max);
}
// Triangle::intersect up to `if t <= delta_t` (triangle.rs:441-545).  Returns true when the reference proceeds past it.
PH_DEV bool tri_test(const RayState& r, f3 p0, f3 p1, f3 p2, float& t_out, float& b0_out, float& b1_out, float& b2_out) {
    f3 o = mk3(r.ox, r.oy, r.oz);
    const int kx = r.kx(), ky = r.ky(), kz = r.kz();
    f3 p0t = permute(p0 - o, kx, ky, kz);
    f3 p1t = permute(p1 - o, kx, ky, kz);
    f3 p2t = permute(p2 - o, kx, ky, kz);
    p0t.x += r.sx * p0t.z; p0t.y += r.sy * p0t.z;
    p1t.x += r.sx * p1t.z; p1t.y += r.sy * p1t.z;
    p2t.x += r.sx * p2t.z; p2t.y += r.sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // f64 fallback at edges (:483-495)
        double a = (double)p2t.x * (double)p1t.y, b = (double)p2t.y * (double)p1t.x;
        e0 = (float)(b - a);
        a = (double)p0t.x * (double)p2t.y; b = (double)p0t.y * (double)p2t.x;
        e1 = (float)(b - a);
        a = (double)p1t.x * (double)p0t.y; b = (double)p1t.y * (double)p0t.x;
        e2 = (float)(b - a);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= r.sz; p1t.z *= r.sz; p2t.z *= r.sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < r.t_max * det)) return false;
    else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > r.t_max * det)) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det, t = t_scaled * inv_det;
    float max_z_t = max_component(vabs(mk3(p0t.z, p1t.z, p2t.z)));
    float delta_z = kGamma3 * max_z_t;
    float max_x_t = max_component(vabs(mk3(p0t.x, p1t.x, p2t.x)));
    float max_y_t = max_component(vabs(mk3(p0t.y, p1t.y, p2t.y)));
    float delta_x = kGamma5 * (max_x_t + max_z_t);
    float delta_y = kGamma5 * (max_y_t + max_z_t);
    float delta_e = 2.0f * (kGamma2 * max_x_t * max_y_t + delta_y * max_x_t + delta_x * max_y_t);
    float max_e = max_component(vabs(mk3(e0, e1, e2)));
    float delta_t = 3.0f * (kGamma3 * max_e * max_z_t + delta_e * max_z_t + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    t_out = t; b0_out = b0; b1_out = b1; b2_out = b2;
    return true;
}

// The barycentrics of tri_test alone (its e0 .. e2, det and inv_det, operation for operation): a kernel that does not carry an accepted hit's b0 .. b2 through the walk
// (traverse_kernel without instancing: five registers less) recomputes them from the winning triangle when the ray retires.  No rejection test is repeated — at that point
// r.t_max is the hit's own t, and `t_scaled > t_max * det` could reject it by a rounding of the product.
PH_DEV void tri_bary(const RayState& r, f3 p0, f3 p1, f3 p2, float& b0_out, float& b1_out, float& b2_out) {
    f3 o = mk3(r.ox, r.oy, r.oz);
    const int kx = r.kx(), ky = r.ky(), kz = r.kz();
    f3 p0t = permute(p0 - o, kx, ky, kz);
    f3 p1t = permute(p1 - o, kx, ky, kz);
    f3 p2t = permute(p2 - o, kx, ky, kz);
    p0t.x += r.sx * p0t.z; p0t.y += r.sy * p0t.z;
    p1t.x += r.sx * p1t.z; p1t.y += r.sy * p1t.z;
    p2t.x += r.sx * p2t.z; p2t.y += r.sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double a = (double)p2t.x * (double)p1t.y, b = (double)p2t.y * (double)p1t.x;
        e0 = (float)(b - a);
        a = (double)p0t.x * (double)p2t.y; b = (double)p0t.y * (double)p2t.x;
        e1 = (float)(b - a);
        a = (double)p1t.x * (double)p0t.y; b = (double)p1t.y * (double)p0t.x;
        e2 = (float)(b - a);
    }
    const float det = e0 + e1 + e2;
    const float inv_det = 1.0f / det;
    b0_out = e0 * inv_det; b1_out = e1 * inv_det; b2_out = e2 * inv_det;
}

// -DPH_PHASE_CLOCK=1 (a measurement build, never shipped or timed): every wave clocks the phases of its loop with s_memtime and adds cycles, executions and active lanes per phase to
// TravParams::phase through LDS tallies (no long-lived registers: the kernel's occupancy stays what it is).  Phases: 0 refill (queue pull, ray load, set-up, root test), 1 node steps,
// 2 instance entry, 3 triangle test, 4 alpha-mask test, 5 instance exit, 6 retire, 7 the whole loop, 8 stack pops, 9 the leaf step as a whole (record fetch + 2 / 3 / 4).
#ifndef PH_PHASE_CLOCK
#define PH_PHASE_CLOCK 0
#endif
#include "phase_clock.h"
// COUNT = true adds per-ray work counters (roofline bookkeeping, never used in a timed run).  For closest-hit rays the
// reference's "nodes visited" is exactly 1 + 2 * (interior nodes whose box test passed): it fetches and tests both children
// of every such node (the far one when it is popped), and nothing else.
//
// Loop shape (wave64): ONE loop, every lane advances its own ray by at most one interior-node step per iteration; lanes that
// have reached a leaf wait until at least PH_LEAF_MIN lanes are at leaves (or no lane has node work left), then all of them
// test ONE triangle each.  A nested "all lanes walk until everybody is at a leaf" (while-while) loop measured 15 % VALU lane
// utilisation on incoherent rays at wave64 (profiles/r01_v1_*); this shape keeps the node code at >80 % and only runs the leaf
// code when a quarter of the wave needs it.  Finished lanes are refilled from a wave-local batch of PH_BATCH consecutive rays
// (one global atomic per batch, not per refill).
#ifndef PH_LEAF_MIN
#define PH_LEAF_MIN 20
#endif
#ifndef PH_REFILL_MIN
#define PH_REFILL_MIN 12
#endif
#ifndef PH_BATCH
#define PH_BATCH 64
#endif
#ifndef PH_LEAF_STEPS
#define PH_LEAF_STEPS 1   // triangles a lane may test per leaf step (a leaf holds up to max_prims_in_node of them)
#endif
// INST = true adds object instancing (TransformedPrimitive): a leaf record may name an instance; the lane then carries its ray
// into instance space, walks the object's aggregate above its current stack height and returns to the scene-level leaf where it
// left it (same order of primitive tests as the reference's recursion).  Compiled separately so scenes without instances keep
// the leaner kernel.
// MIXED = true serves both ray kinds of one wavefront round from ONE launch: indices [0, n_cl) are closest-hit rays, [n_cl, n_cl + n_sh)
// any-hit rays (each lane knows its kind).  Every launch ends with a latency-bound tail (the last rays' dependent loads, ~0.4 ms
// whatever the launch size), so one launch per round instead of two removes one tail per bounce.
// ALPHA != 0 adds alpha-mask textures (triangle.rs:587-607 / 868-898): a candidate hit on a mesh whose alpha or shadowalpha is a texture is
// accepted only if the texture, evaluated at the hit's uv / p with no differentials, is not 0.  Its own instantiations, so every other scene keeps the leaner kernels.
//   ALPHA = 1: every alpha texture of the scene is made of image maps under the uv mapping, constants, scale and mix — what scene files use for cut-outs.  Without
//              differentials an image map is one bilinear look-up on the finest level whatever its filter (mipmap/mod.rs:222-231, :250-262), so the whole test is a few
//              dozen instructions and is inlined (alpha_accept_lean, texture.h).
//   ALPHA = 2: any texture class: an out-of-line call of the general evaluator (alpha_accept).  The callee's register appetite becomes the kernel's (336 VGPRs, one wave
//              per SIMD: measured 30 x slower on a foliage scene than ALPHA = 1) — correct, and kept for the procedural masks only.
static __device__ __noinline__ bool alpha_accept(const DeviceScene* dsc, uint32_t tri_index, float b0, float b1, float b2, uint32_t any_hit);
static __device__ __forceinline__ void alpha_general_prepare();
static __device__ __forceinline__ bool alpha_accept_lean(const DeviceScene& sc, uint32_t prim, uint32_t mesh, float b0, float b1, float b2, bool any_hit);
// ALPHA_MIN > 0 (round 4) makes the alpha-mask test a PHASE of its own.  The test is a chain of dependent fetches (mesh record -> vertex indices -> uv -> texture program -> MIPMap
//   record -> four texels): on configs[4] it ran in 85 % of the leaf steps for 3.2 lanes on average and took a third of the waves' cycles (scripts/phase_clock.py,
//   profiles/r04_phase_clock_config4_*).  A lane whose candidate hit needs the mask's verdict now stays at its record with `alpha_wait` set; once ALPHA_MIN lanes of the wave wait (or
//   the wave has no interior-node work left) they fetch their records again, repeat the triangle test — same ray, same t_max, same numbers; cheaper than carrying t and the
//   barycentrics in registers this kernel does not have — and evaluate their masks together.  The order of tests along a ray is unchanged, so the result is.
// WPE > 0 compiles the kernel for exactly that many waves per SIMD (= resident 256-thread blocks per CU): the register allocator then fits the budget
// (7: 72 VGPRs, 8: 64) instead of taking what it likes; 0 leaves the choice to the compiler (same code as before).
template <bool ANYHIT, bool COUNT = false, int LEAF_MIN = PH_LEAF_MIN, int REFILL_MIN = PH_REFILL_MIN, int LDS_DEPTH = PH_LDS_DEPTH, int NODE_STEPS = 1, bool INST = false, bool MIXED = false,
          int ALPHA = 0, int WPE = 0, int ALPHA_MIN = 0>
__global__ __launch_bounds__(PH_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : 1, WPE ? WPE : 8))) void traverse_kernel(DeviceScene sc, TravParams p) {
    __shared__ uint2 lds_stack[LDS_DEPTH][PH_TRAV_BLOCK];
    // INST: the scene-level ray's origin and what ray_setup derived from it (six IEEE divides), parked while the lane walks an instance: leaving an instance is then nine LDS reads instead of
    // a reload of the ray and a second ray_setup.  The direction is not parked — the scene-level one stays in three registers of its own (re-reading it from the ray queue at every instance cost 4 %:
    // by then the ray's line has left the caches) —, the signs come back from the reciprocals and kz waits in the top bits of in_inst: 9 KB per block instead of 13, so that five blocks of (11-entry stack + this) fit a CU.
    __shared__ float inst_save[INST ? 9 : 1][INST ? PH_TRAV_BLOCK : 1];
    const uint32_t tid = threadIdx.x;
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __shared__ unsigned long long phc_lds[36];
    if (tid < 36) phc_lds[tid] = 0ull;
    __syncthreads();
#endif
    PHC_BEGIN(7);
    if (ALPHA == 2) alpha_general_prepare();   // (the general evaluator's Perlin table, into LDS: texture.h)
    const uint32_t lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * PH_TRAV_BLOCK + tid;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t n_first = p.n_ptr ? *p.n_ptr : p.n;
    const uint32_t n_rays = MIXED ? n_first + *p.n2_ptr : n_first;
    bool ah = ANYHIT;                        // this lane's ray kind (MIXED: per ray, else the template's)

    bool has_ray = false;
    uint32_t batch_next = 0, batch_end = 0;  // wave-uniform
    bool exhausted = false;                  // wave-uniform: the global queue has nothing left
    uint32_t head_cur = p.n_heads > 1u ? blockIdx.x % p.n_heads : 0u, heads_dry = 0u;  // wave-uniform: the head this wave pulls from, heads found empty
    uint32_t ray_index = 0;
    RayState r;
    uint32_t cur = PH_INVALID_REF;           // interior node index, or PH_LEAF_BIT | index of the NEXT TriRec to test
    int sp = 0;
    // LEAN (no instancing): an accepted hit is remembered by its TriRec alone (hit_tri, none = all ones); primitive id, material class and barycentrics are read / recomputed
    // from it when the ray retires (tri_bary) — five registers the walk does not carry.  Not with instancing: a hit inside an instance would have to carry the scene-level ray into
    // that instance's space again at the end, and that second transform + set-up in the kernel costs more registers than it frees (102 -> 115, measured).
    constexpr bool LEAN = !INST;
    uint32_t hit_prim = 0xFFFFFFFFu, hit_tri = LEAN ? 0xFFFFFFFFu : 0u, hit_cls = 0u;
    float hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    bool occluded = false;
    uint32_t c_nodes[2] = {0, 0}, c_tris[2] = {0, 0}, c_rays[2] = {0, 0}, c_visits = 0, c_culled = 0;   // c_culled: stack entries dropped at pop time (their box lies behind a hit found since)  // COUNT: [0] this launch's first kind, [1] MIXED any-hit
    // instancing state (INST only)
    float wdx = 0.0f, wdy = 0.0f, wdz = 0.0f;   // the scene-level ray's direction: every instance it meets transforms it anew (RayState does not keep directions)
    uint32_t in_inst = 0;                 // instance number + 1 while inside an object's aggregate
    int inst_sp = 0;                      // stack height at entry: the object's entries live above it
    float world_tmax = 0.0f;              // the scene-level ray's t_max at entry
    uint32_t cont_ref = PH_INVALID_REF;   // next record of the scene-level leaf, or PH_INVALID_REF = pop
    bool inst_hit = false;                // a primitive of the current instance was accepted
    uint32_t hit_inst = 0;
    bool alpha_wait = false;              // ALPHA_MIN > 0: the triangle at `cur` passed the geometric test and waits for its alpha mask's verdict

    auto push = [&](uint32_t ref, float tmin) {
        uint2 e = make_uint2(ref, __float_as_uint(tmin));
        if (sp < LDS_DEPTH) lds_stack[sp][tid] = e;
        else if (sp < (INST ? 2 * PH_MAX_STACK : PH_MAX_STACK)) p.spill[(size_t)(sp - LDS_DEPTH) * p.total_threads + gtid] = e;  // INST: scene + object entries share the stack
        else { *p.error_flag = 1u; return; }
        sp++;
    };
    // Pops are DEFERRED (round 3): where the walk needs the next stack entry it only notes so (cur = PH_NEED_POP, or PH_INVALID_REF at once when the stack is empty), and every node step
    // begins with ONE pop attempt for the lanes that owe one — an entry that fails `t_min < ray.t_max` leaves the lane owing the next.  The entries are popped in the same order and
    // tested against the same t_max (a lane that owes a pop is at no leaf, so nothing shrinks its t_max meanwhile): same visits, same bits.  The pop loop with its two address spaces was
    // expanded inline at three places: this form issues a third fewer scalar instructions and branches per frame (SQ_INSTS_SALU 2.60e11 -> 1.71e11 on configs[2]).  Worth 5.5 % on the
    // instancing kernel (1 180 -> 1 115 ms of traversal per frame on 1 000 x 10 k instances), nothing on the flat one (703.5 against 705.7 ms; gpurun r03an) — HISTORY §4, "what bounds the kernel".
    auto owe_pop = [&]() -> uint32_t {
        const int floor_sp = (INST && in_inst) ? inst_sp : 0;
        return sp > floor_sp ? PH_NEED_POP : PH_INVALID_REF;
    };
    auto pop_once = [&]() {
        const int floor_sp = (INST && in_inst) ? inst_sp : 0;
        if (sp > floor_sp) {
            sp--;
            // two loads in two address spaces, kept apart by (empty, distinct) asm statements: merged into one load through a generic pointer, the
            // LDS-aperture test fails instruction selection on this toolchain (ROCm 7.2, gfx950: "Operand has incorrect register class", V_CMP_NE_U32 against src_shared_base)
            uint2 e;
            if (sp < LDS_DEPTH) { e = lds_stack[sp][tid]; asm volatile("; stack entry from LDS" : "+v"(e.x), "+v"(e.y)); }
            else { e = p.spill[(size_t)(sp - LDS_DEPTH) * p.total_threads + gtid]; asm volatile("; stack entry from the spill region" : "+v"(e.x), "+v"(e.y)); }
            if (COUNT && ah) c_visits++;  // the reference fetches and box-tests every node it pops
            if (COUNT && !(__uint_as_float(e.y) < r.t_max)) c_culled++;
            cur = (__uint_as_float(e.y) < r.t_max) ? e.x : PH_NEED_POP;
        } else cur = PH_INVALID_REF;
    };

    // A finished ray (cur == PH_INVALID_REF outside any instance) is written out when the wave next refills — all the lanes that finished since the last refill together (round 4;
    // rounds 1 - 3 retired a ray in the pass it finished in: 4.2 lanes per execution of this code on configs[2], 5.4 % of the waves' cycles, scripts/phase_clock.py).  Until then the
    // lane idles exactly as an empty one does: every test of the loop below excludes cur == PH_INVALID_REF.
    auto retire = [&]() {
            PHC_BEGIN(6);
            if (MIXED && ah) p.out2[ray_index - n_first] = occluded ? 1 : 0;
            else if (!MIXED && ANYHIT) reinterpret_cast<uint8_t*>(p.out)[ray_index] = occluded ? 1 : 0;
            else {
                float4* hp = reinterpret_cast<float4*>(reinterpret_cast<HitOut*>(p.out) + ray_index);
                if (LEAN) {
                    if (hit_tri != 0xFFFFFFFFu) {
                        const float4* tp = reinterpret_cast<const float4*>(sc.tris + hit_tri);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        tri_bary(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), hb0, hb1, hb2);
                        hit_prim = __float_as_uint(a.w); hit_cls = (__float_as_uint(b.w) >> PH_TRI_CLASS_SHIFT) & PH_TRI_KEY_MASK;
                    } else { hit_prim = 0xFFFFFFFFu; hit_tri = 0u; hit_cls = 0u; hb0 = hb1 = hb2 = 0.0f; }
                }
                PH_STREAM_STORE(make_float4(r.t_max, __uint_as_float(hit_prim), hb0, hb1), hp);
                PH_STREAM_STORE(make_float4(hb2, __uint_as_float(hit_tri), __uint_as_float(INST ? hit_inst : 0u), __uint_as_float(hit_cls)), hp + 1);  // pad[0] = the hit's TriRec, pad[1] = instance + 1, pad[2] = material class | material id << 3
            }
            has_ray = false;
            if (COUNT) c_rays[(MIXED && ah) ? 1 : 0]++;
            PHC_END(6);
    };

    for (;;) {
        // ---- refill idle lanes from the wave's batch; grab a new batch with one atomic when it runs dry ---------------------------
        {
            PHC_BEGIN(0);
            const bool finished = has_ray && cur == PH_INVALID_REF && !(INST && in_inst);
            const uint64_t idle = __ballot(!has_ray || finished);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle >= (uint32_t)REFILL_MIN || n_idle == 64u || (n_idle && __ballot(has_ray && !finished) == 0ull)) {
                if (finished) retire();
                if (batch_next == batch_end && !exhausted) {
                    if (p.n_heads > 1u) {  // one attempt per pass of the outer loop: a drained head sends the wave to the next one
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(p.heads + 16u * head_cur, p.batch);
                        b = __shfl(b, 0);
                        // position b of head h's chunk list -> queue position
                        const uint32_t chunk = b / p.head_chunk;
                        const uint64_t q = ((uint64_t)chunk * p.n_heads + head_cur) * p.head_chunk + (b - chunk * p.head_chunk);
                        if (q < (uint64_t)n_rays) { batch_next = (uint32_t)q; batch_end = ((uint32_t)q + p.batch < n_rays) ? (uint32_t)q + p.batch : n_rays; }
                        else {
                            heads_dry++; head_cur = head_cur + 1u == p.n_heads ? 0u : head_cur + 1u;
                            if (heads_dry >= p.n_heads) exhausted = true;
                        }
                    } else {
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(p.counter, p.batch);
                        b = __shfl(b, 0);
                        if (b >= n_rays) exhausted = true;
                        else { batch_next = b; batch_end = (b + p.batch < n_rays) ? b + p.batch : n_rays; }
                    }
                }
                const uint32_t avail = batch_end - batch_next;
                if (avail) {
                    const uint32_t rank = (uint32_t)__popcll(idle & lane_lt);
                    if (!has_ray && rank < avail) {
                        ray_index = p.order ? PH_STREAM_LOAD(p.order + batch_next + rank) : batch_next + rank;
                        if (MIXED) ah = ray_index >= n_first;
                        const float4* rp = reinterpret_cast<const float4*>((MIXED && ah) ? p.rays2 + (ray_index - n_first) : p.rays + ray_index);
                        const float4 a = PH_STREAM_LOAD(rp), b = PH_STREAM_LOAD(rp + 1);
                        RayIn in; in.ox = a.x; in.oy = a.y; in.oz = a.z; in.t_max = a.w; in.dx = b.x; in.dy = b.y; in.dz = b.z; in.time = b.w;
                        ray_setup(r, in);
                        has_ray = true; sp = 0; occluded = false; alpha_wait = false;
                        if (LEAN) hit_tri = 0xFFFFFFFFu; else { hit_prim = 0xFFFFFFFFu; hit_tri = 0u; hb0 = hb1 = hb2 = 0.0f; }
                        if (INST) { in_inst = 0; hit_inst = 0; wdx = in.dx; wdy = in.dy; wdz = in.dz; }
                        // root: the reference tests nodes[0].bounds first (bvh/mod.rs:189-190)
                        cur = PH_INVALID_REF;
                        if (COUNT && ah) c_visits++;
                        if (sc.root_ref != PH_INVALID_REF) {
                            float tmin;
                            const bool h = box_test(r, r.nx() ? sc.root_hi[0] : sc.root_lo[0], r.nx() ? sc.root_lo[0] : sc.root_hi[0],
                                                    r.ny() ? sc.root_hi[1] : sc.root_lo[1], r.ny() ? sc.root_lo[1] : sc.root_hi[1],
                                                    r.nz() ? sc.root_hi[2] : sc.root_lo[2], r.nz() ? sc.root_lo[2] : sc.root_hi[2], tmin);
                            if (h && tmin < r.t_max) cur = sc.root_ref;
                        }
                    }
                    batch_next += (n_idle < avail) ? n_idle : avail;
                }
                PHC_END(0);
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted && batch_next == batch_end) break;
            continue;
        }

        // ---- NODE_STEPS interior-node steps for every lane that is at an interior node ---------------------------------------------------
#pragma unroll
        for (int step = 0; step < NODE_STEPS; step++) {
        if (has_ray && cur == PH_NEED_POP) { PHC_BEGIN(8); pop_once(); PHC_END(8); }
        if (has_ray && !(cur & PH_LEAF_BIT)) {   // (PH_INVALID_REF and PH_NEED_POP have the leaf bit set)
            PHC_BEGIN(1);
            const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
            const float4 q0 = np[0], q1 = np[1], q2 = np[2];
            const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
            if (COUNT) c_nodes[(MIXED && ah) ? 1 : 0]++;
            // q0 = x0[0],x0[1],y0[0],y0[1]; q1 = z0[0],z0[1],x1[0],x1[1]; q2 = y1[0],y1[1],z1[0],z1[1]
            float t0, t1;
#if PH_NODE_PK
            ph_mask m0, m1;
            node_boxes(r, q0, q1, q2, m0, t0, m1, t1);
            // bvh/mod.rs:206-214: dir_is_neg[axis] -> second child first
#if defined(__HIP_DEVICE_COMPILE__)
            const ph_mask m_neg = lanes(__builtin_amdgcn_ubfe(r.sgn, q3.z, 1u) != 0u);
#else
            const ph_mask m_neg = lanes(((r.sgn >> q3.z) & 1u) != 0u);
#endif
            const bool neg_axis = lane_of(m_neg);
            const uint32_t near_ref = neg_axis ? q3.y : q3.x, far_ref = neg_axis ? q3.x : q3.y;
            const bool near_hit = lane_of((m_neg & m1) | (~m_neg & m0)), far_hit = lane_of((m_neg & m0) | (~m_neg & m1));
            const float far_t = neg_axis ? t0 : t1;
#else
            bool h0, h1;
            h0 = box_test(r, r.nx() ? q0.y : q0.x, r.nx() ? q0.x : q0.y, r.ny() ? q0.w : q0.z, r.ny() ? q0.z : q0.w,
                               r.nz() ? q1.y : q1.x, r.nz() ? q1.x : q1.y, t0);
            h1 = box_test(r, r.nx() ? q1.w : q1.z, r.nx() ? q1.z : q1.w, r.ny() ? q2.y : q2.x, r.ny() ? q2.x : q2.y,
                               r.nz() ? q2.w : q2.z, r.nz() ? q2.z : q2.w, t1);
            h0 = h0 & (t0 < r.t_max);
            h1 = h1 & (t1 < r.t_max);
            const int neg_axis = (int)((r.sgn >> q3.z) & 1u);
            // bvh/mod.rs:206-214: dir_is_neg[axis] -> second child first
            const uint32_t near_ref = neg_axis ? q3.y : q3.x, far_ref = neg_axis ? q3.x : q3.y;
            const bool near_hit = neg_axis ? h1 : h0, far_hit = neg_axis ? h0 : h1;
            const float far_t = neg_axis ? t0 : t1;
#endif
            if (COUNT && ah) {
                // any-hit rays stop early, so "nodes the reference visits" is counted as it goes: the near child always, the far child when
                // it is popped — also when its box test fails, hence the NaN-keyed entry (pop() counts it and skips it)
                if (near_hit) { c_visits++; if (!far_hit) push(far_ref, __builtin_nanf("")); }
                else c_visits += 2;
            }
            if (near_hit) { cur = near_ref; if (far_hit) push(far_ref, far_t); }
            else if (far_hit) cur = far_ref;
            else cur = owe_pop();
            PHC_END(1);
        }
        }

        // ---- leaf work: one triangle per lane, once enough lanes are waiting at leaves ----------------------------------------------------
        {
            // ALPHA_MIN > 0: lanes that wait for an alpha-mask verdict stand at their record too, but only join a leaf step that fires FOR THEM (enough of them wait)
            const bool waiting = ALPHA && ALPHA_MIN > 0 && has_ray && alpha_wait;
            const bool at_leaf = has_ray && cur < PH_NEED_POP && (cur & PH_LEAF_BIT) && !waiting;
            const uint64_t lm = __ballot(at_leaf), am = (ALPHA && ALPHA_MIN > 0) ? __ballot(waiting) : 0ull;
            if ((lm | am) != 0ull) {
                const uint64_t nm = __ballot(has_ray && (cur == PH_NEED_POP || !(cur & PH_LEAF_BIT)));
                const bool fire_leaf = lm != 0ull && ((uint32_t)__popcll(lm) >= (uint32_t)LEAF_MIN || nm == 0ull || exhausted);  // queue drained: no throughput left to protect, only the tail's latency
                const bool fire_alpha = am != 0ull && ((uint32_t)__popcll(am) >= (uint32_t)(ALPHA_MIN > 0 ? ALPHA_MIN : 1) || nm == 0ull || exhausted);
                if (fire_leaf || fire_alpha) {
#pragma unroll
                    for (int ls = 0; ls < PH_LEAF_STEPS; ls++)
                    // (the lane's state as it is NOW: with more than one triangle per step a lane may have left its leaf, or begun to wait, in the step before; the waiting lanes join the first only)
                    if ((has_ray && cur < PH_NEED_POP && (cur & PH_LEAF_BIT) && !(ALPHA && ALPHA_MIN > 0 && alpha_wait) && (ls == 0 ? at_leaf : true) && fire_leaf) || (ls == 0 && waiting && fire_alpha)) {
                        PHC_BEGIN(9);
                        const uint32_t ti = INST ? (cur & ~(PH_LEAF_BIT | PH_LEAF_INST_HINT)) : (cur & ~PH_LEAF_BIT);
                        const float4* tp = reinterpret_cast<const float4*>(sc.tris + ti);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        // INST: a reference that carries the hint names an instance record: its transform is fetched with it (scene_types.h, PH_LEAF_INST_HINT)
                        const bool hinted = INST && (cur & PH_LEAF_INST_HINT) != 0u;
                        float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0, e2 = e0;
                        if (INST && hinted) { const float4* ep = reinterpret_cast<const float4*>(sc.inst_extra) + 3u * (size_t)ti; e0 = ep[0]; e1 = ep[1]; e2 = ep[2]; }
                        const uint32_t flags = __float_as_uint(b.w);
                        bool last = (flags & PH_TRI_LAST) != 0;
                        // the next record of this leaf (with its own hint)
                        const uint32_t next_ref = INST ? (((cur & ~PH_LEAF_INST_HINT) + 1u) | ((flags & PH_TRI_NEXT_INST) ? PH_LEAF_INST_HINT : 0u)) : cur + 1u;
                        float t, b0, b1, b2;
                        if (INST && (flags & PH_TRI_INSTANCE)) {
                            PHC_BEGIN(2);
                            // TransformedPrimitive::intersect / intersect_p (transformed_primitive.rs:51-73).  The record itself holds the object's bounds, root and flags (filled in at upload)
                            if (!hinted) { const float4* ep = reinterpret_cast<const float4*>(sc.inst_extra) + 3u * (size_t)ti; e0 = ep[0]; e1 = ep[1]; e2 = ep[2]; }   // no hint (e.g. a root that is a leaf): one more round trip, same numbers
                            const uint32_t iflags = __float_as_uint(c.y), iroot = __float_as_uint(c.x);
                            float w2i[16] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w, 0.0f, 0.0f, 0.0f, 1.0f};
                            if (iflags & PH_INST_GENERAL) { const InstRec& I = sc.instances[__float_as_uint(a.w)]; for (int k = 12; k < 16; k++) w2i[k] = I.w2i[k]; }   // a projective last row: from the InstRec
                            world_tmax = r.t_max; cont_ref = last ? PH_INVALID_REF : next_ref;
                            in_inst = (__float_as_uint(a.w) + 1u) | ((r.sgn >> 3) << 30); inst_sp = sp; inst_hit = false;   // bits 30..31: the scene-level ray's kz
                            inst_save[0][tid] = r.ox; inst_save[1][tid] = r.oy; inst_save[2][tid] = r.oz;
                            inst_save[3][tid] = r.ix; inst_save[4][tid] = r.iy; inst_save[5][tid] = r.iz; inst_save[6][tid] = r.sx; inst_save[7][tid] = r.sy; inst_save[8][tid] = r.sz;
                            // (round 3, measured and not kept: proving a miss of the object's bound with three hardware reciprocals before paying this set-up — the proof rarely
                            //  succeeds once the instance's world bound has been passed, 86.6 -> 93.2 ms on 1 000 x 10 k instances — and postponing the triangle half of ray_setup
                            //  to the first triangle met inside: no difference, gpurun r03g)
                            const RayIn in = xf_ray(w2i, r, mk3(wdx, wdy, wdz), 0.0f);
                            ray_setup(r, in);
                            cur = PH_INVALID_REF;
                            if (iflags & PH_INST_SINGLE) cur = iroot;  // the lone primitive itself, no aggregate
                            else {
                                float tmin;   // the record's p0 / p1 = the object's root bounds
                                const bool h = box_test(r, r.nx() ? b.x : a.x, r.nx() ? a.x : b.x, r.ny() ? b.y : a.y, r.ny() ? a.y : b.y,
                                                        r.nz() ? b.z : a.z, r.nz() ? a.z : b.z, tmin);
                                if (h && tmin < r.t_max) cur = iroot;
                            }
                            PHC_END(2);
                        } else {
                        PHC_BEGIN(3);
                        if (COUNT) c_tris[(MIXED && ah) ? 1 : 0]++;
                        const bool second_meeting = ALPHA && ALPHA_MIN > 0 && alpha_wait;   // the lane waited at this record for its alpha mask's verdict: the test below is the one it already passed
                        alpha_wait = false;
                        if (tri_test(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), t, b0, b1, b2)) {
                            // post-t rejections: degenerate triangle (triangle.rs:567-570 / 862-866), alpha == 0 (:603 / :886-893)
                            const uint32_t reject = ah ? (PH_TRI_BOGUS | PH_TRI_ALPHA0 | PH_TRI_SALPHA0) : (PH_TRI_BOGUS | PH_TRI_ALPHA0);
                            bool accept = !(flags & reject);
                            if (ALPHA && accept && (flags & PH_TRI_ALPHATEX)) {
                                if (ALPHA_MIN > 0 && !second_meeting) { alpha_wait = true; accept = false; }   // first meeting: the lane stays at this record and waits for companions
                                else {   // (second meeting: same ray, same t_max, same numbers — now followed by the mask's verdict)
                                PHC_BEGIN(4);
                                accept = ALPHA == 1 ? alpha_accept_lean(sc, __float_as_uint(a.w), __float_as_uint(c.w), b0, b1, b2, ah) : alpha_accept(sc.self, ti, b0, b1, b2, ah ? 1u : 0u);
                                PHC_END(4);
                                }
                            }
                            if (accept) {
                                if (ah) occluded = true;
                                else {
                                    r.t_max = t; hit_tri = ti;
                                    if (!LEAN) { hit_prim = __float_as_uint(a.w); hb0 = b0; hb1 = b1; hb2 = b2; hit_cls = (flags >> PH_TRI_CLASS_SHIFT) & PH_TRI_KEY_MASK; }
                                    if (INST) { hit_inst = in_inst & 0x3FFFFFFFu; inst_hit = true; }
                                }
                            }
                        }
                        if (ALPHA && ALPHA_MIN > 0 && alpha_wait) { }   // (not advanced)
                        else if (ah && occluded) cur = PH_INVALID_REF;
                        else if (last) cur = owe_pop();
                        else cur = next_ref;
                        PHC_END(3);
                        }
                        PHC_END(9);
                    }
                }
            }
        }

        // ---- leave an exhausted instance: back to the scene-level ray, `r.t_max = ray.t_max` only if something was hit inside ----------
        if (INST && has_ray && in_inst && cur == PH_INVALID_REF) {
            PHC_BEGIN(5);
            if (!(ah && occluded)) {
                const float t_new = inst_hit ? r.t_max : world_tmax;
                r.ox = inst_save[0][tid]; r.oy = inst_save[1][tid]; r.oz = inst_save[2][tid];
                r.ix = inst_save[3][tid]; r.iy = inst_save[4][tid]; r.iz = inst_save[5][tid]; r.sx = inst_save[6][tid]; r.sy = inst_save[7][tid]; r.sz = inst_save[8][tid];
                r.sgn = (r.ix < 0.0f ? 1u : 0u) | (r.iy < 0.0f ? 2u : 0u) | (r.iz < 0.0f ? 4u : 0u) | ((in_inst >> 30) << 3);
                r.t_max = t_new;
                in_inst = 0;
                cur = (cont_ref != PH_INVALID_REF) ? cont_ref : owe_pop();
            } else in_inst = 0;
            PHC_END(5);
        }

    }
    PHC_END(7);
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
    if (tid < 36 && p.phase) atomicAdd(p.phase + tid, phc_lds[tid]);
#endif
    if (COUNT) {
        // p.counts: closest-hit {nodes, tris, rays} any-hit {nodes, tris, rays} any-hit reference node visits
        const int k0 = (!MIXED && ANYHIT) ? 3 : 0;
        atomicAdd(p.counts + k0 + 0, (unsigned long long)c_nodes[0]);
        atomicAdd(p.counts + k0 + 1, (unsigned long long)c_tris[0]);
        atomicAdd(p.counts + k0 + 2, (unsigned long long)c_rays[0]);
        if (MIXED) {
            atomicAdd(p.counts + 3, (unsigned long long)c_nodes[1]);
            atomicAdd(p.counts + 4, (unsigned long long)c_tris[1]);
            atomicAdd(p.counts + 5, (unsigned long long)c_rays[1]);
        }
        atomicAdd(p.counts + 6, (unsigned long long)c_visits);
        atomicAdd(p.counts + 7, (unsigned long long)c_culled);
    }
}

}  // namespace ph
