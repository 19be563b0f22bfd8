// gfx950 device math in the reference's exact f32 expression order.  Built with -ffp-contract=off, IEEE divide/sqrt
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved: every + - * / sqrt below rounds exactly
// like the reference's Rust f32 code, so traversal results are bit-identical to a CPU run of the same expressions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PH_DEV __device__ __forceinline__

namespace ph {

// core/src/pbrt/common.rs:19-49
constexpr float kInf = __builtin_huge_valf();
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kPiOver2 = kPi * 0.5f;
constexpr float kPiOver4 = kPi * 0.25f;
constexpr float kTwoPi = kPi * 2.0f;
constexpr float kInvTwoPi = 1.0f / kTwoPi;
constexpr float kMachEps = 5.9604644775390625e-08f;  // f32::EPSILON * 0.5
constexpr float kShadowEps = 0.0001f;
constexpr float kOneMinusEps = 0x1.fffffep-1f;
// gamma(n) = n*eps / (1 - n*eps)  (core/src/pbrt/common.rs:132-134), evaluated in f32 like the reference
constexpr float gamma_c(int n) { return ((float)n * kMachEps) / (1.0f - (float)n * kMachEps); }
constexpr float kGamma2 = gamma_c(2), kGamma3 = gamma_c(3), kGamma5 = gamma_c(5), kGamma6 = gamma_c(6), kGamma7 = gamma_c(7);
constexpr float kBoxScale = 1.0f + 2.0f * kGamma3;  // bounds3.rs:303-305

// generic abs/min/max of core/src/pbrt/common.rs:66-107 (written with < and >: NaN and -0.0 behave like the reference)
PH_DEV float pabs(float n) { return n < 0.0f ? -n : n; }
PH_DEV float pminf(float a, float b) { return a < b ? a : b; }
PH_DEV float pmaxf(float a, float b) { return a > b ? a : b; }
PH_DEV float pclampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
PH_DEV int pmini(int a, int b) { return a < b ? a : b; }
PH_DEV int pmaxi(int a, int b) { return a > b ? a : b; }

// Outlined IEEE divide / sqrt / f64 transcendentals for the big shade kernel (PH_OUTLINE_MATH): hipcc expands every correctly
// rounded f32 divide into ~12 instructions and every f64 sin/cos into a few hundred; inlined at ~120 / ~10 sites they pushed the
// shade kernel past the instruction cache (64 KB of code, ~37 cycles per issued instruction measured).  Arguments and results
// travel in registers, so a call costs a handful of cycles.  The traversal kernels keep the inline forms.
#ifdef PH_OUTLINE_MATH
__device__ __noinline__ float ph_div(float a, float b) { return a / b; }
__device__ __noinline__ float ph_sqrt(float a) { return sqrtf(a); }
#else
PH_DEV float ph_div(float a, float b) { return a / b; }
PH_DEV float ph_sqrt(float a) { return sqrtf(a); }
#endif

struct f3 { float x, y, z; };
PH_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PH_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PH_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PH_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PH_DEV f3 operator*(f3 a, float f) { return mk3(a.x * f, a.y * f, a.z * f); }
PH_DEV f3 operator*(float f, f3 a) { return mk3(a.x * f, a.y * f, a.z * f); }
PH_DEV f3 operator/(f3 a, float f) { float inv = ph_div(1.0f, f); return mk3(inv * a.x, inv * a.y, inv * a.z); }  // vector3.rs:406-418
PH_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PH_DEV float abs_dot(f3 a, f3 b) { return pabs(dot(a, b)); }
PH_DEV f3 cross(f3 a, f3 b) { return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }
PH_DEV float length_squared(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PH_DEV float length(f3 a) { return ph_sqrt(length_squared(a)); }
PH_DEV f3 normalize(f3 a) { return a / length(a); }
PH_DEV f3 vabs(f3 a) { return mk3(pabs(a.x), pabs(a.y), pabs(a.z)); }
PH_DEV float max_component(f3 a) { return a.x > a.y ? (a.x > a.z ? a.x : a.z) : (a.y > a.z ? a.y : a.z); }  // vector3.rs:93-107
PH_DEV int max_dimension(f3 a) { return a.x > a.y ? (a.x > a.z ? 0 : 2) : (a.y > a.z ? 1 : 2); }          // vector3.rs:109-123
PH_DEV float comp(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
PH_DEV f3 permute(f3 a, int kx, int ky, int kz) { return mk3(comp(a, kx), comp(a, ky), comp(a, kz)); }
PH_DEV f3 face_forward(f3 n, f3 v) { return dot(n, v) < 0.0f ? -n : n; }
PH_DEV float distance_squared(f3 a, f3 b) { return length_squared(a - b); }
PH_DEV void coordinate_system(f3 v1, f3& v2, f3& v3) {  // core/src/geometry/coordinate_system.rs:12-20
    if (pabs(v1.x) > pabs(v1.y)) v2 = mk3(-v1.z, 0.0f, v1.x) / ph_sqrt(v1.x * v1.x + v1.z * v1.z);
    else v2 = mk3(0.0f, v1.z, -v1.y) / ph_sqrt(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
PH_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// Transcendentals: evaluated in f64 and rounded once.  The reference calls the platform's f32 libm, which is not
// correctly rounded (glibc sinf differs from the rounded f64 value for ~1.3 % of arguments, atan2f for ~16 %), so no
// device routine can match it bit for bit; the rounded-f64 value is within 1 ulp of it and is reproducible on any host.
#ifdef PH_OUTLINE_MATH
#define PH_TRIG __device__ __noinline__
#else
#define PH_TRIG PH_DEV
#endif
struct sc2 { float s, c; };
PH_TRIG float d_sin(float x) { return (float)sin((double)x); }
PH_TRIG float d_cos(float x) { return (float)cos((double)x); }
PH_TRIG sc2 d_sincos2(float x) { double ds, dc; sincos((double)x, &ds, &dc); sc2 r; r.s = (float)ds; r.c = (float)dc; return r; }
PH_DEV void d_sincos(float x, float& s, float& c) { sc2 r = d_sincos2(x); s = r.s; c = r.c; }
PH_TRIG float d_acos(float x) { return (float)acos((double)x); }
PH_TRIG float d_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

PH_DEV float spherical_theta(f3 v) { return d_acos(pclampf(v.z, -1.0f, 1.0f)); }                    // geometry/util.rs:41-44
PH_DEV float spherical_phi(f3 v) { float p = d_atan2(v.y, v.x); return p < 0.0f ? p + kTwoPi : p; }  // :46-56

PH_DEV float next_float_up(float v) {  // core/src/pbrt/common.rs:205-222
    if (__builtin_isinf(v) && v > 0.0f) return v;
    float nv = (v == -0.0f) ? 0.0f : v;
    uint32_t ui = __float_as_uint(nv);
    if (nv >= 0.0f) ui += 1; else ui -= 1;
    return __uint_as_float(ui);
}
PH_DEV float next_float_down(float v) {  // :227-243
    if (__builtin_isinf(v) && v < 0.0f) return v;
    float nv = (v == 0.0f) ? -0.0f : v;
    uint32_t ui = __float_as_uint(nv);
    if (nv > 0.0f) ui -= 1; else ui += 1;
    return __uint_as_float(ui);
}
// Rust `as i32` / `as usize` float casts saturate and map NaN to 0
PH_DEV int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
PH_DEV uint32_t f2u_sat(float f) {  // for values known to be small non-negative indices
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// RGBSpectrum (core/src/spectrum/rgb_spectrum.rs)
struct spec { float r, g, b; };
PH_DEV spec mks(float r, float g, float b) { spec s; s.r = r; s.g = g; s.b = b; return s; }
PH_DEV spec mks1(float v) { return mks(v, v, v); }
PH_DEV spec operator+(spec a, spec b) { return mks(a.r + b.r, a.g + b.g, a.b + b.b); }
PH_DEV spec operator*(spec a, spec b) { return mks(a.r * b.r, a.g * b.g, a.b * b.b); }
PH_DEV spec operator*(spec a, float f) { return mks(a.r * f, a.g * f, a.b * f); }
PH_DEV spec operator*(float f, spec a) { return a * f; }
PH_DEV spec operator/(spec a, float f) { return a * ph_div(1.0f, f); }  // rgb_spectrum.rs:255-263
PH_DEV bool is_black(spec a) { return a.r == 0.0f && a.g == 0.0f && a.b == 0.0f; }
PH_DEV float lum_y(spec a) { return 0.212671f * a.r + 0.715160f * a.g + 0.072169f * a.b; }
PH_DEV float max_component_value(spec a) { return pmaxf(pmaxf(a.r, a.g), a.b); }
PH_DEV bool has_nans(spec a) { return a.r != a.r || a.g != a.g || a.b != a.b; }

}  // namespace ph
