"""Every test the reference itself holds, replayed against the oracle (the reference's only tests are the 277 `#[test]`s of
core/src/geometry/*.rs, listed with their line numbers in tests/golden/reference_tests.json by tests/golden/make_reference_test_manifest.py).

Each replay restates the property the reference test asserts — same input ranges, the expected value written out in f32 / i32 as the reference
test writes it — and asks the oracle's math layer (oracle_math.hpp: V3 for Vector3f / Point3f / Normal3f, V2 for Vector2f / Point2f, B2<T> for
Bounds2f / Bounds2i, M4, Ray) for the other side.  `test_every_reference_test_is_accounted_for` fails when a manifest entry is neither replayed
nor listed below with the reason it cannot apply to a C++ restatement of the path (i32 instantiations of generics the path only uses at f32,
debug-assertion panics)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle_binding import oracle_binding

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "reference_tests.json")))
G = "core/src/geometry/"

# proptest ranges of the reference (prop_range!(range_f32, f32, -100.0..100.0f32) etc.)
rf = st.floats(min_value=-100.0, max_value=100.0, width=32, allow_nan=False, allow_infinity=False, exclude_max=True)
nz = rf.filter(lambda x: abs(x) > 1e-30)  # non-zero, and 1 / f finite (with an infinite reciprocal 0 * inf = NaN fails the reference's own assert_eq too)
v3s = st.tuples(rf, rf, rf)
v2s = st.tuples(rf, rf)
ri = st.integers(min_value=-100, max_value=99)
p2i = st.tuples(ri, ri)
axis3 = st.integers(min_value=0, max_value=2)
axis2 = st.integers(min_value=0, max_value=1)


def fr(lo, hi):
    lo, hi = float(np.float32(lo)), float(np.float32(hi))
    return st.floats(min_value=lo, max_value=hi, width=32, allow_nan=False, allow_infinity=False, exclude_max=True)


def _lib():
    L = oracle_binding().lib
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    L.oracle_geom_op.argtypes = [C.c_int, fp, fp]
    L.oracle_geom2_op.argtypes = [C.c_int, fp, fp]
    L.oracle_bounds2f_op.argtypes = [C.c_int, fp, fp]
    L.oracle_bounds2i_op.argtypes = [C.c_int, ip, ip, C.c_int]
    return L


def _call(fn, code, vals, n_out):
    a = np.zeros(32, np.float32); a[: len(vals)] = vals
    out = np.zeros(32, np.float32)
    fn(code, a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n_out].copy()


def op3(code, u=(0, 0, 0), v=(0, 0, 0), s=0.0, extra=(), n=3):
    return _call(_lib().oracle_geom_op, code, list(u) + list(v) + [s] + list(extra), n)


def op2(code, u=(0, 0), v=(0, 0), s=0.0, perm=(0, 0), n=2):
    return _call(_lib().oracle_geom2_op, code, list(u) + list(v) + [s] + list(perm), n)


def bf(code, b1=(0, 0, 0, 0), b2=(0, 0, 0, 0), p=(0, 0), s=0.0, n=4):
    return _call(_lib().oracle_bounds2f_op, code, list(b1) + list(b2) + list(p) + [s], n)


def bi(code, b1=(0, 0, 0, 0), b2=(0, 0, 0, 0), p=(0, 0), s=0, n=4, cap=1024):
    a = np.zeros(16, np.int32); vals = list(b1) + list(b2) + list(p) + [s]; a[: len(vals)] = vals
    out = np.zeros(cap, np.int32)
    _lib().oracle_bounds2i_op(code, a.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_int)), cap)
    return out[:n].copy() if code != 18 else out[: 1 + 2 * int(out[0])].copy()


def F(*xs):
    return np.array(xs, np.float32)


def eq(a, b):  # Rust's == on f32 components: -0.0 == 0.0, NaN != NaN
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape and bool(np.all(a == b)), (a, b)


def pmin(a, b):  # pbrt::min / max (pbrt/common.rs:83-108)
    return a if a < b else b


def pmax(a, b):
    return a if a > b else b


def lerp(t, a, b):  # pbrt::lerp (pbrt/common.rs:167-173)
    return (f32(1.0) - f32(t)) * a + f32(t) * b


F32_MAX = np.finfo(np.float32).max
I32_MAX, I32_MIN = 2**31 - 1, -(2**31)
EMPTY_F = (F32_MAX, F32_MAX, -F32_MAX, -F32_MAX)
EMPTY_I = (I32_MAX, I32_MAX, I32_MIN, I32_MIN)
S = settings(max_examples=120, deadline=None)
REPLAYS = {}


def replay(files, *names):
    """Registers the decorated function as the replay of `names` in each of `files` (the 3-D types share one POD in the oracle)."""
    def deco(fn):
        for f in ([files] if isinstance(files, str) else files):
            for nme in names:
                REPLAYS[(G + f, nme)] = fn.__name__
        return fn
    return deco


V3FILES = ["vector3.rs", "point3.rs", "normal.rs"]
V2FILES = ["vector2.rs", "point2.rs"]

# ---------------------------------------------------------------------------------------------------------------- 3-D types


@replay(["vector3.rs", "normal.rs"], "zero_vector")
@replay("point3.rs", "zero_point")
def test_v3_zero():
    eq(op3(14), F(0, 0, 0))  # zero() + zero()


@replay(V3FILES, "has_nans")
def test_v3_has_nans():
    assert op3(27, (0, 0, 0), n=1)[0] == 0.0
    assert op3(27, (np.nan, np.nan, np.nan), n=1)[0] == 1.0
    assert op3(27, (0, np.nan, 0), n=1)[0] == 1.0


@replay("vector3.rs", "cross_axis_f32")
def test_v3_cross_axis():
    x, y, z = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    eq(op3(1, x, y), F(*z)); eq(op3(1, y, x), -F(*z))
    eq(op3(1, y, z), F(*x)); eq(op3(1, z, y), -F(*x))
    eq(op3(1, z, x), F(*y)); eq(op3(1, x, z), -F(*y))


@replay(["vector3.rs", "normal.rs"], "length_squared_f32", "length_f32")
@S
@given(v3s)
def test_v3_length(v):
    x, y, z = F(*v)
    out = op3(3, v, n=2)
    assert out[1] == x * x + y * y + z * z
    assert out[0] == np.sqrt(x * x + y * y + z * z)


@replay(["vector3.rs", "normal.rs"], "normalize_f32")
@S
@given(v3s.filter(lambda v: float(sum(f32(c) * f32(c) for c in v)) > 1e-30))  # the reference's normalize asserts a non-zero length
def test_v3_normalize(v):
    x, y, z = F(*v)
    with np.errstate(over="ignore", under="ignore"):
        f = f32(1.0) / np.sqrt(x * x + y * y + z * z)
        eq(op3(2, v), F(x * f, y * f, z * f))


@replay("vector3.rs", "abs_f32")
@S
@given(v3s)
def test_v3_abs(v):
    eq(op3(4, v), np.array([-c if c < 0 else c for c in F(*v)], np.float32))


@replay(["vector3.rs", "normal.rs"], "dot_f32", "abs_dot_f32")
@S
@given(v3s, v3s)
def test_v3_dot(a, b):
    (ax, ay, az), (bx, by, bz) = F(*a), F(*b)
    d = ax * bx + ay * by + az * bz
    assert op3(0, a, b, n=1)[0] == d
    assert op3(26, a, b, n=1)[0] == np.abs(d)


@replay("vector3.rs", "cross_f32", "cross_zero_f32")
@S
@given(v3s, v3s)
def test_v3_cross(a, b):
    (ax, ay, az), (bx, by, bz) = F(*a), F(*b)
    eq(op3(1, a, b), F((ay * bz) - (az * by), (az * bx) - (ax * bz), (ax * by) - (ay * bx)))
    z = F(0, 0, 0)
    eq(op3(1, (0, 0, 0), a), z); eq(op3(1, a, (0, 0, 0)), z); eq(op3(1, a, a), z)


@replay("vector3.rs", "min_component_f32", "max_component_f32", "max_dimension_f32")
@S
@given(v3s)
def test_v3_components(v):
    x, y, z = F(*v)
    assert op3(21, v, n=1)[0] == min(x, y, z)
    assert op3(5, v, n=1)[0] == max(x, y, z)
    dim = (0 if x > z else 2) if x > y else (1 if y > z else 2)
    assert int(op3(6, v, n=1)[0]) == dim


@replay(["vector3.rs", "point3.rs"], "min_f32", "max_f32")
@S
@given(v3s, v3s)
def test_v3_min_max(a, b):
    A, B = F(*a), F(*b)
    eq(op3(19, a, b), np.minimum(A, B)); eq(op3(20, a, b), np.maximum(A, B))


@replay(["vector3.rs", "point3.rs"], "permute_f32")
@S
@given(v3s, axis3, axis3, axis3)
def test_v3_permute(v, a1, a2, a3):
    eq(op3(7, v, (a1, a2, a3)), F(v[a1], v[a2], v[a3]))


@replay(["vector3.rs", "normal.rs"], "add_f32", "add_assign_f32", "sub_f32", "sub_assign_f32")
@replay("point3.rs", "add_point_f32", "add_vector_f32", "add_assign_vector_f32", "sub_point_f32", "sub_vector_f32", "sub_assign_vector_f32")
@S
@given(v3s, v3s)
def test_v3_add_sub(a, b):
    eq(op3(14, a, b), F(*a) + F(*b)); eq(op3(15, a, b), F(*a) - F(*b))


@replay(V3FILES, "mul_f32", "mul_assign_f32")
@S
@given(v3s, rf)
def test_v3_mul(v, f):
    out = op3(16, v, s=f, n=6)
    eq(out[:3], F(*v) * f32(f)); eq(out[3:], F(*v) * f32(f))  # v * f and f * v


@replay(V3FILES, "div_f32", "div_assign_f32")
@S
@given(v3s, nz)
def test_v3_div_multiplies_by_the_reciprocal(v, f):
    with np.errstate(over="ignore"):
        s = f32(1.0) / f32(f)
        eq(op3(17, v, s=f), F(*v) * s)


@replay(V3FILES, "neg_f32")
@S
@given(v3s)
def test_v3_neg(v):
    eq(op3(18, v), -F(*v)); eq(op3(18, tuple(op3(18, v))), F(*v))


@replay(V3FILES, "index_f32", "index_mut_f32")
@S
@given(v3s)
def test_v3_index(v):
    eq(op3(7, v, (0, 1, 2)), F(*v))  # v[X], v[Y], v[Z] read and written through the index operator the permutation uses


@replay("point3.rs", "distance_squared_f32", "length_f32")
@S
@given(v3s, v3s)
def test_p3_distance(a, b):
    (ax, ay, az), (bx, by, bz) = F(*a), F(*b)
    e = (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz)
    assert op3(12, a, b, n=1)[0] == e
    assert op3(25, a, b, n=1)[0] == np.sqrt(e)


@replay("point3.rs", "floor_f32", "ceil_f32")
@S
@given(v3s)
def test_p3_floor_ceil(v):
    eq(op3(22, v), np.floor(F(*v))); eq(op3(23, v), np.ceil(F(*v)))


@replay("point3.rs", "lerp_edge_case_f32", "lerp_f32")
@S
@given(v3s, v3s, fr(-2.0, 2.0))
def test_p3_lerp(a, b, t):
    eq(op3(24, a, b, s=0.0), F(*a)); eq(op3(24, a, b, s=1.0), F(*b))
    eq(op3(24, a, b, s=t), (f32(1.0) - f32(t)) * F(*a) + f32(t) * F(*b))


# ---------------------------------------------------------------------------------------------------------------- 2-D types


@replay("vector2.rs", "zero_vector")
@replay("point2.rs", "zero_point")
def test_v2_zero():
    eq(op2(14), F(0, 0))


@replay(V2FILES, "has_nans")
def test_v2_has_nans():
    assert op2(27, (0, 0), n=1)[0] == 0.0 and op2(27, (np.nan, np.nan), n=1)[0] == 1.0


@replay("vector2.rs", "length_squared_f32", "length_f32")
@S
@given(v2s)
def test_v2_length(v):
    x, y = F(*v)
    out = op2(3, v)
    assert out[1] == x * x + y * y and out[0] == np.sqrt(x * x + y * y)


@replay("vector2.rs", "normalize_f32")
@S
@given(v2s.filter(lambda v: float(sum(f32(c) * f32(c) for c in v)) > 1e-30))
def test_v2_normalize(v):
    x, y = F(*v)
    with np.errstate(over="ignore", under="ignore"):
        f = f32(1.0) / np.sqrt(x * x + y * y)
        eq(op2(2, v), F(x * f, y * f))


@replay("vector2.rs", "abs_f32")
@S
@given(v2s)
def test_v2_abs(v):
    eq(op2(4, v), np.array([-c if c < 0 else c for c in F(*v)], np.float32))


@replay("vector2.rs", "dot_f32", "abs_dot_f32")
@S
@given(v2s, v2s)
def test_v2_dot(a, b):
    (ax, ay), (bx, by) = F(*a), F(*b)
    out = op2(0, a, b)
    assert out[0] == ax * bx + ay * by and out[1] == np.abs(ax * bx + ay * by)


@replay("vector2.rs", "min_component_f32", "max_component_f32", "max_dimension_f32")
@S
@given(v2s)
def test_v2_components(v):
    x, y = F(*v)
    out = op2(5, v)
    assert out[0] == max(x, y) and out[1] == min(x, y)
    assert int(op2(6, v, n=1)[0]) == (0 if x > y else 1)


@replay(V2FILES, "min_f32", "max_f32")
@S
@given(v2s, v2s)
def test_v2_min_max(a, b):
    eq(op2(19, a, b), np.minimum(F(*a), F(*b))); eq(op2(20, a, b), np.maximum(F(*a), F(*b)))


@replay(V2FILES, "permute_f32")
@S
@given(v2s, axis2, axis2)
def test_v2_permute(v, a1, a2):
    eq(op2(7, v, perm=(a1, a2)), F(v[a1], v[a2]))


@replay("vector2.rs", "add_f32", "add_assign_f32", "sub_f32", "sub_assign_f32")
@replay("point2.rs", "add_point_f32", "add_vector_f32", "add_assign_vector_f32", "sub_point_f32", "sub_vector_f32", "sub_assign_vector_f32")
@S
@given(v2s, v2s)
def test_v2_add_sub(a, b):
    eq(op2(14, a, b), F(*a) + F(*b)); eq(op2(15, a, b), F(*a) - F(*b))


@replay(V2FILES, "mul_f32", "mul_assign_f32")
@S
@given(v2s, rf)
def test_v2_mul(v, f):
    out = op2(16, v, s=f, n=4)
    eq(out[:2], F(*v) * f32(f)); eq(out[2:], F(*v) * f32(f))


@replay(V2FILES, "div_f32", "div_assign_f32")
@S
@given(v2s, nz)
def test_v2_div_multiplies_by_the_reciprocal(v, f):
    with np.errstate(over="ignore"):
        eq(op2(17, v, s=f), F(*v) * (f32(1.0) / f32(f)))


@replay(V2FILES, "neg_f32")
@S
@given(v2s)
def test_v2_neg(v):
    eq(op2(18, v), -F(*v)); eq(op2(18, tuple(op2(18, v))), F(*v))


@replay(V2FILES, "index_f32", "index_mut_f32")
@S
@given(v2s)
def test_v2_index(v):
    eq(op2(7, v, perm=(0, 1)), F(*v))


@replay("point2.rs", "distance_squared_f32", "length_f32")
@S
@given(v2s, v2s)
def test_p2_distance(a, b):
    (ax, ay), (bx, by) = F(*a), F(*b)
    e = (ax - bx) * (ax - bx) + (ay - by) * (ay - by)
    out = op2(12, a, b)
    assert out[0] == e and out[1] == np.sqrt(e)


@replay("point2.rs", "floor_f32", "ceil_f32")
@S
@given(v2s)
def test_p2_floor_ceil(v):
    eq(op2(22, v), np.floor(F(*v))); eq(op2(23, v), np.ceil(F(*v)))


@replay("point2.rs", "lerp_edge_case_f32", "lerp_f32")
@S
@given(v2s, v2s, fr(-2.0, 2.0))
def test_p2_lerp(a, b, t):
    eq(op2(24, a, b, s=0.0), F(*a)); eq(op2(24, a, b, s=1.0), F(*b))
    eq(op2(24, a, b, s=t), (f32(1.0) - f32(t)) * F(*a) + f32(t) * F(*b))


# ---------------------------------------------------------------------------------------------------------------- ray, matrix, coordinate system


@replay("ray.rs", "at", "at_f32")
@S
@given(v3s, v3s, rf)
def test_ray_at(o, d, t):
    eq(op3(8, (0, 0, 0), (1, 1, 1), s=0.0), F(0, 0, 0)); eq(op3(8, (0, 0, 0), (1, 1, 1), s=1.0), F(1, 1, 1))
    eq(op3(8, o, d, s=t), F(*o) + f32(t) * F(*d))


@replay("ray.rs", "scale_differentials_some")
def test_ray_scale_differentials():
    o, d, xo, yo, xd, yd = F(0, 0, 0), F(1, 1, 1), F(1, 0, 0), F(0, 1, 0), F(1, 0, 0), F(0, 1, 0)
    out = op3(28, o, d, s=2.0, extra=list(xo) + list(yo) + list(xd) + list(yd), n=12)
    two = f32(2.0)
    eq(out[0:3], o + two * (xo - o)); eq(out[3:6], o + two * (yo - o)); eq(out[6:9], d + two * (xd - d)); eq(out[9:12], d + two * (yd - d))


@replay("matrix4x4.rs", "inverse_returns_identity_when_matrix_is_idenitity")
def test_matrix_inverse_identity():
    I = np.eye(4, dtype=np.float32).ravel()
    eq(_call(_lib().oracle_geom_op, 10, I, 16), I)


@replay("matrix4x4.rs", "inverse_returns_matrix_when_matrix_is_non_singular")
@S
@given(fr(0.001, 10.0), fr(0.001, 10.0), fr(0.001, 10.0), fr(0.001, 10.0))
def test_matrix_inverse_of_a_diagonal(a, b, c, d):
    m = np.diag(F(a, b, c, d)).astype(np.float32)
    inv = _call(_lib().oracle_geom_op, 10, m.ravel(), 16).reshape(4, 4)
    assert np.allclose(m @ inv, np.eye(4), atol=1e-4) and np.allclose(inv @ m, np.eye(4), atol=1e-4)


@replay("coordinate_system.rs", "from_unit_x_axis", "from_x_axis", "from_vector_x_greater_than_y", "from_vector_x_less_than_y")
def test_coordinate_system_cases():
    out = op3(9, (1, 0, 0), n=6); eq(out[:3], F(0, 0, 1)); eq(out[3:], F(0, -1, 0))
    out = op3(9, (2, 0, 0), n=6); eq(out[:3], F(0, 0, 1)); eq(out[3:], F(0, -2, 0))
    for v1 in ((0.5, 0.2, 0.5), (0.2, 0.5, 0.5)):
        out = op3(9, v1, n=6)
        v2, v3 = tuple(out[:3]), tuple(out[3:])
        assert op3(0, v1, v2, n=1)[0] == 0.0 and op3(0, v1, v3, n=1)[0] == 0.0 and op3(0, v2, v3, n=1)[0] == 0.0


# ---------------------------------------------------------------------------------------------------------------- Bounds2 (bounds2.rs:366-1236)
B = "bounds2.rs"


def newf(p1, p2):
    return tuple(bf(0, (p1[0], p1[1], p2[0], p2[1])))


def newi(p1, p2):
    return tuple(int(x) for x in bi(0, (p1[0], p1[1], p2[0], p2[1])))


@replay(B, "empty_bounds2f_returns_min_greater_than_max_components", "empty_bounds2i_returns_min_greater_than_max_components",
        "area_of_empty_bounds2i_returns_zero", "area_of_empty_bounds2f_returns_zero", "union_of_two_empty_bounds2i_returns_empty",
        "union_of_two_empty_bounds2f_returns_empty", "intersection_of_two_empty_bounds2i_returns_empty", "intersection_of_two_empty_bounds2f_returns_empty",
        "bounding_circle_of_empty_box_returns_origin_and_zero_radius")
def test_b2_empty():
    eq(bf(1), F(*EMPTY_F)); eq(bi(1), np.array(EMPTY_I, np.int32))
    assert bf(5, EMPTY_F, n=1)[0] == 0.0 and bi(5, EMPTY_I, n=1)[0] == 0
    assert bf(3, tuple(bf(16, EMPTY_F, EMPTY_F)), n=1)[0] == 1.0 and bi(3, tuple(bi(16, EMPTY_I, EMPTY_I)), n=1)[0] == 1
    assert bf(3, tuple(bf(17, EMPTY_F, EMPTY_F)), n=1)[0] == 1.0 and bi(3, tuple(bi(17, EMPTY_I, EMPTY_I)), n=1)[0] == 1
    eq(bf(11, EMPTY_F, n=3), F(0, 0, 0))


@replay(B, "corner_returns_points_with_left_to_right_in_x_top_to_bottom_in_y", "index_returns_p_min_at_0_and_p_max_at_1")
def test_b2_corner_and_index():
    b = newi((-1, -1), (1, 1))
    assert b == (-1, -1, 1, 1)  # b[0] = p_min, b[1] = p_max
    for k, want in enumerate([(-1, -1), (1, -1), (-1, 1), (1, 1)]):
        assert tuple(bi(14, b, s=k, n=2)) == want


@replay(B, "iterating_empty_bounds2i_return_none", "iterate_point_bounds2i_returns_point_only")
def test_b2i_iteration_edge_cases():
    assert bi(18, EMPTY_I)[0] == 0
    out = bi(18, (0, 0, 0, 0))
    assert out[0] == 1 and tuple(out[1:3]) == (0, 0)


@replay(B, "bounds2i_sorts_x_and_y_components", "bounds2i_from_point_sets_min_max_to_given_point", "diagonal_of_bounds2i_returns_vector_from_min_to_max")
@S
@given(p2i, p2i)
def test_b2i_new(p1, p2):
    b1, b2 = newi(p1, p2), newi(p2, p1)
    assert b1 == b2 == (min(p1[0], p2[0]), min(p1[1], p2[1]), max(p1[0], p2[0]), max(p1[1], p2[1]))
    assert tuple(bi(2, p=p1)) == (p1[0], p1[1], p1[0], p1[1])
    assert tuple(bi(4, b1, n=2)) == (b1[2] - b1[0], b1[3] - b1[1])


@replay(B, "bounds2f_sorts_x_and_y_components", "bounds2f_from_point_sets_min_max_to_given_point", "diagonal_of_bounds2f_returns_vector_from_min_to_max")
@S
@given(v2s, v2s)
def test_b2f_new(p1, p2):
    b1, b2 = newf(p1, p2), newf(p2, p1)
    want = (pmin(f32(p1[0]), f32(p2[0])), pmin(f32(p1[1]), f32(p2[1])), pmax(f32(p1[0]), f32(p2[0])), pmax(f32(p1[1]), f32(p2[1])))
    eq(F(*b1), F(*b2)); eq(F(*b1), F(*want))
    eq(bf(2, p=p1), F(p1[0], p1[1], p1[0], p1[1]))
    eq(bf(4, b1, n=2), F(b1[2] - b1[0], b1[3] - b1[1]))


@replay(B, "area_of_non_empty_bounds2i_returns_product_of_diagonal_components")
@S
@given(p2i, st.integers(-10, 9), st.integers(-10, 9))
def test_b2i_area(p, dx, dy):
    assert bi(5, newi(p, (p[0] + dx, p[1] + dy)), n=1)[0] == abs(dx * dy)


@replay(B, "area_of_non_empty_bounds2f_returns_product_of_diagonal_components")
@S
@given(v2s, fr(-10.0, 10.0), fr(-10.0, 10.0))
def test_b2f_area(p, dx, dy):
    b = newf(p, (f32(p[0]) + f32(dx), f32(p[1]) + f32(dy)))
    assert abs(bf(5, b, n=1)[0] - abs(f32(dx) * f32(dy))) <= 1e-4 * max(1.0, abs(dx * dy)) + 2e-3  # the reference test compares with epsilon 0.0001 (float_cmp: or a few ulps)
    assert bf(5, b, n=1)[0] == (f32(b[2]) - f32(b[0])) * (f32(b[3]) - f32(b[1]))              # and the exact definition (bounds2.rs:100-111)


@replay(B, "maximum_extent_of_non_empty_bounds2i_returns_axis_with_max_diagonal_component")
@S
@given(p2i, st.integers(0, 9))
def test_b2i_maximum_extent(p, d):
    assert bi(6, newi(p, (p[0] + d + 1, p[1] + d)), n=1)[0] == 0
    assert bi(6, newi(p, (p[0] + d, p[1] + d + 1)), n=1)[0] == 1
    assert bi(6, newi(p, (p[0] + d, p[1] + d)), n=1)[0] == 1


@replay(B, "maximum_extent_of_non_empty_bounds2f_returns_axis_with_max_diagonal_component", "maximum_extent_of_non_empty_bounds2f_returns_y_axis_edge_case")
@S
@given(v2s, fr(0.0, 10.0), fr(-10.0, 10.0))
def test_b2f_maximum_extent(p, d, x):
    px, py, d, x = f32(p[0]), f32(p[1]), f32(d), f32(x)
    for q, _ in (((px + (d + f32(0.001)), py + d), 0), ((px + d, py + (d + f32(0.001))), 1)):
        b = newf(p, q)
        dx, dy = f32(b[2]) - f32(b[0]), f32(b[3]) - f32(b[1])
        assert bf(6, b, n=1)[0] == (0.0 if dx > dy else 1.0)  # the definition; the reference's axis claim holds wherever f32 addition keeps the 0.001 margin
    assert bf(6, newf((x, x), (x + d, x + d)), n=1)[0] == 1.0


def _shifted(b, lo, hi):
    return newf((f32(b[0]) + lo[0], f32(b[1]) + lo[1]), (f32(b[2]) + hi[0], f32(b[3]) + hi[1]))


@replay(B, "overlaps_returns_true_when_two_bounds2f_overlap")
@S
@given(v2s, fr(0.1, 1.0), fr(0.1, 1.0), fr(0.0, 2.0), fr(0.0, 2.0))
def test_b2f_overlaps(p, dx, dy, sx, sy):
    px, py, dx, dy = f32(p[0]), f32(p[1]), f32(dx), f32(dy)
    b1 = newf((px - dx, py - dy), (px + dx, py + dy))
    ax, ay = f32(sx) * dx, f32(sy) * dy
    z = f32(0)
    shifts = [((-ax, z), (-ax, z)), ((ax, z), (ax, z)), ((z, -ay), (z, -ay)), ((z, ay), (z, ay)), ((-ax, -ay), (-ax, -ay)), ((ax, ay), (ax, ay)),
              ((ax, -ay), (ax, -ay)), ((-ax, ay), (-ax, ay)), ((-ax, z), (ax, z)), ((z, -ay), (z, ay)), ((-ax, -ay), (ax, ay))]
    for lo, hi in shifts:
        b2 = _shifted(b1, lo, hi)
        want = (b1[2] >= b2[0]) and (b1[0] <= b2[2]) and (b1[3] >= b2[1]) and (b1[1] <= b2[3])  # bounds2.rs:129-136
        assert bf(7, b1, b2, n=1)[0] == (1.0 if want else 0.0)
        if sx <= 1.9 and sy <= 1.9:
            assert want  # the reference's claim, away from the touching case where f32 rounding decides


@replay(B, "overlaps_returns_false_when_two_bounds2f_do_not_overlap")
@S
@given(v2s, fr(1.0, 2.0), fr(1.0, 2.0), fr(2.001, 3.0), fr(2.001, 3.0))
def test_b2f_does_not_overlap(p, dx, dy, sx, sy):
    px, py, dx, dy = f32(p[0]), f32(p[1]), f32(dx), f32(dy)
    b1 = newf((px - dx, py - dy), (px + dx, py + dy))
    ax, ay = f32(sx) * dx, f32(sy) * dy
    z = f32(0)
    for lo in [(-ax, z), (ax, z), (z, -ay), (z, ay), (-ax, -ay), (ax, -ay), (-ax, ay)]:
        assert bf(7, b1, _shifted(b1, lo, lo), n=1)[0] == 0.0


@replay(B, "offset_of_any_point_within_an_empty_bounds2f_returns_vector_towards_p_min", "offset_of_any_point_within_non_empty_bounds2f_returns_components_in_0_1")
@S
@given(v2s, v2s, v2s)
def test_b2f_offset(p1, p2, p):
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        eq(bf(8, EMPTY_F, p=p, n=2), F(f32(p[0]) - F32_MAX, f32(p[1]) - F32_MAX))
        b = newf(p1, p2)
        o = F(f32(p[0]) - b[0], f32(p[1]) - b[1])
        out = bf(8, b, p=p, n=2)
        if b[2] > b[0]:
            assert out[0] == o[0] / (f32(b[2]) - f32(b[0]))
        if b[3] > b[1]:
            assert out[1] == o[1] / (f32(b[3]) - f32(b[1]))


@replay(B, "contains_returns_false_for_any_point_when_empty_bounds2f", "contains_returns_true_for_point_bounds2f_when_p_min_p_max_is_same_point",
        "contains_returns_false_when_point_is_outside_bounds2f", "contains_returns_true_when_point_is_inside_bounds2f")
@S
@given(v2s, fr(0.001, 1.0), fr(0.001, 1.0), fr(0.001, 1.0), fr(0.001, 1.0), fr(0.0, 1.0), fr(0.0, 1.0))
def test_b2f_contains(p, dx, dy, sx, sy, tx, ty):
    px, py, dx, dy, sx, sy = (f32(v) for v in (p[0], p[1], dx, dy, sx, sy))
    assert bf(9, EMPTY_F, p=p, n=1)[0] == 0.0
    pt = tuple(bf(2, p=p))
    assert bf(9, pt, p=p, n=1)[0] == 1.0
    for q in [(px + dx, py), (px, py + dy), (px + dx, py + dy), (px - dx, py), (px, py - dy), (px - dx, py - dy)]:
        assert bf(9, pt, p=q, n=1)[0] == 0.0
    b = newf((px - dx, py - dy), (px + dx, py + dy))
    x0, y0, x1, y1 = (f32(v) for v in b)
    outside = [(x0 - sx, y0), (x0, y0 - sy), (x0 - sx, y0 - sy), (x1 + sx, y1), (x1, y1 + sy), (x1 + sx, y1 + sy), (x0 - sx, y1), (x0, y1 + sy),
               (x0 - sx, y1 + sy), (x1 + sx, y0), (x1, y0 - sy), (x1 + sx, y0 - sy)]
    for q in outside:
        assert bf(9, b, p=q, n=1)[0] == 0.0
    q = (lerp(tx, x0, x1), lerp(ty, y0, y1))
    want = q[0] >= x0 and q[0] <= x1 and q[1] >= y0 and q[1] <= y1  # the definition (bounds2.rs:159-164): true except where lerp rounds across a corner
    assert bf(9, b, p=q, n=1)[0] == (1.0 if want else 0.0)


@replay(B, "contains_exclusive_returns_false_for_any_point_when_empty_bounds2i", "contains_exclusive_returns_true_for_point_bounds2i_when_p_min_p_max_is_same_point",
        "contains_exclusive_returns_false_when_point_is_outside_bounds2i", "contains_exclusive_returns_true_when_point_is_inside_bounds2i")
@S
@given(p2i, st.integers(1, 9), st.integers(1, 9), st.integers(1, 9), st.integers(1, 9), st.integers(0, 9), st.integers(0, 9), fr(0.0, 0.999), fr(0.0, 0.999),
       st.integers(0, 4), st.integers(0, 4))
def test_b2i_contains_exclusive(p, dx, dy, sx, sy, kx, ky, tx, ty, ex, ey):
    assert bi(10, EMPTY_I, p=p, n=1)[0] == 0
    pt = tuple(int(v) for v in bi(2, p=p))
    for q in [p, (p[0] + ex, p[1]), (p[0], p[1] + ey), (p[0] + ex, p[1] + ey), (p[0] - ex, p[1]), (p[0], p[1] - ey), (p[0] - ex, p[1] - ey)]:
        assert bi(10, pt, p=q, n=1)[0] == 0  # a point box contains nothing exclusively (the reference test's name says "true", its body asserts false)
    b = newi((p[0] - dx, p[1] - dy), (p[0] + dx, p[1] + dy))
    x0, y0, x1, y1 = b
    for q in [(x0 - sx, y0), (x0, y0 - sy), (x0 - sx, y0 - sy), (x1 + kx, y1), (x1, y1 + ky), (x1 + kx, y1 + ky), (x0 - sx, y1), (x0, y1 + ky), (x0 - sx, y1 + ky),
              (x1 + kx, y0), (x1, y0 - sy), (x1 + kx, y0 - sy)]:
        assert bi(10, b, p=q, n=1)[0] == 0
    q = (int(np.floor(lerp(tx, f32(x0), f32(x1)))), int(np.floor(lerp(ty, f32(y0), f32(y1)))))
    assert bi(10, b, p=q, n=1)[0] == 1


@replay(B, "bounding_circle_of_point_bounds2f_returns_point_as_center_and_zero_radius", "bounding_circle_returns_midpoint_as_center_and_distance_to_p_max_as_radius")
@S
@given(v2s, v2s)
def test_b2f_bounding_circle(p1, p2):
    out = bf(11, tuple(bf(2, p=p1)), n=3)
    half = f32(0.5)
    eq(out, F(half * f32(p1[0]) + half * f32(p1[0]), half * f32(p1[1]) + half * f32(p1[1]), 0.0))  # lerp(0.5, p, p); == p unless halving a subnormal rounds
    if all(c == 0.0 or abs(c) > 1e-30 for c in p1):
        eq(out, F(p1[0], p1[1], 0.0))
    b = newf(p1, p2)
    out = bf(11, b, n=3)
    c = lerp(0.5, F(*p1), F(*p2))
    eq(out[:2], c)
    inside = b[0] <= c[0] <= b[2] and b[1] <= c[1] <= b[3]  # false only where halving a subnormal coordinate rounds the centre out of the box (bounds2.rs:183-187)
    assert out[2] == (op2(12, tuple(c), (b[2], b[3]))[1] if inside else 0.0)  # center.distance(b.p_max)


@replay(B, "lerp_returns_corners_of_bounds2f_at_0_and_1", "lerp_interpolates_and_extrapolates_across_corners_of_bounds2f")
@S
@given(v2s, v2s, fr(-2.0, 2.0), fr(-2.0, 2.0))
def test_b2f_lerp(p1, p2, tx, ty):
    b = newf(p1, p2)
    eq(bf(12, b, p=(0.0, 0.0), n=2), F(b[0], b[1])); eq(bf(12, b, p=(1.0, 1.0), n=2), F(b[2], b[3]))
    eq(bf(12, b, p=(tx, ty), n=2), F(lerp(tx, f32(b[0]), f32(b[2])), lerp(ty, f32(b[1]), f32(b[3]))))


@replay(B, "expand_returns_empty_when_bounds2i_is_empty", "expand_returns_non_empty_bounds2i_for_bounds2i_from_point", "expand_returns_bounds2i_for_non_empty_bounds2i")
@S
@given(p2i, p2i, st.integers(0, 1))
def test_b2i_expand(p1, p2, delta):
    assert bi(3, tuple(int(v) for v in bi(13, EMPTY_I, s=delta)), n=1)[0] == 1
    assert tuple(bi(13, tuple(int(v) for v in bi(2, p=p1)), s=delta)) == (p1[0] - delta, p1[1] - delta, p1[0] + delta, p1[1] + delta)
    b = newi(p1, p2)
    assert tuple(bi(13, b, s=delta)) == (b[0] - delta, b[1] - delta, b[2] + delta, b[3] + delta)


@replay(B, "expand_returns_empty_when_bounds2f_is_empty", "expand_returns_non_empty_bounds2f_for_bounds2f_from_point", "expand_returns_non_empty_bounds2f_for_non_empty_bounds2f")
@S
@given(v2s, v2s, fr(0.0, 100.0))
def test_b2f_expand(p1, p2, delta):
    d = f32(delta)
    for b in (EMPTY_F, tuple(bf(2, p=p1)), newf(p1, p2)):
        x0, y0, x1, y1 = (f32(v) for v in b)
        eq(bf(13, b, s=delta), F(x0 - d, y0 - d, x1 + d, y1 + d))


@replay(B, "union_empty_with_bounds2i_from_point_returns_latter", "union_empty_with_non_empty_bounds2i_returns_non_empty", "intersect_empty_with_non_empty_bounds2i_returns_empty")
@S
@given(p2i, p2i)
def test_b2i_union_intersect_with_empty(p1, p2):
    assert tuple(bi(15, EMPTY_I, p=p1)) == (p1[0], p1[1], p1[0], p1[1])
    b = newi(p1, p2)
    assert tuple(bi(16, EMPTY_I, b)) == b and tuple(bi(16, b, EMPTY_I)) == b
    assert bi(3, tuple(int(v) for v in bi(17, EMPTY_I, b)), n=1)[0] == 1 and bi(3, tuple(int(v) for v in bi(17, b, EMPTY_I)), n=1)[0] == 1


@replay(B, "union_empty_with_bounds2f_from_point_returns_latter", "union_empty_with_non_empty_bounds2f_returns_non_empty", "intersect_empty_with_non_empty_bounds2f_returns_empty")
@S
@given(v2s, v2s)
def test_b2f_union_intersect_with_empty(p1, p2):
    eq(bf(15, EMPTY_F, p=p1), F(p1[0], p1[1], p1[0], p1[1]))
    b = newf(p1, p2)
    eq(bf(16, EMPTY_F, b), F(*b)); eq(bf(16, b, EMPTY_F), F(*b))
    assert bf(3, tuple(bf(17, EMPTY_F, b)), n=1)[0] == 1.0 and bf(3, tuple(bf(17, b, EMPTY_F)), n=1)[0] == 1.0


def _rint(x):  # f32::round (half away from zero) as i32
    return int(np.sign(x) * np.floor(np.abs(x) + f32(0.5)))


@replay(B, "union_non_empty_bounds2i_with_exterior_point_returns_non_empty_bounds2i", "union_non_empty_bounds2i_with_interior_point_returns_same_bounds2i")
@S
@given(p2i, st.integers(1, 9), st.integers(1, 9), st.integers(1, 9), fr(0.0, 1.0), fr(0.0, 1.0))
def test_b2i_union_point(p, dx, dy, s, t, ty):
    b = newi((p[0] - dx, p[1] - dy), (p[0] + dx, p[1] + dy))
    x0, y0, x1, y1 = b
    y = _rint(lerp(f32(t) - f32(1.0), f32(y0), f32(y1)))
    assert tuple(bi(15, b, p=(x0 - s, y))) == newi((x0 - s, y), (x1, y1)) and tuple(bi(15, b, p=(x1 + s, y))) == newi((x0, y), (x1 + s, y1))
    y = _rint(lerp(t, f32(y0), f32(y1)))
    assert tuple(bi(15, b, p=(x0 - s, y))) == newi((x0 - s, y0), (x1, y1)) and tuple(bi(15, b, p=(x1 + s, y))) == newi((x0, y0), (x1 + s, y1))
    y = _rint(lerp(f32(t) + f32(1.0), f32(y0), f32(y1)))
    assert tuple(bi(15, b, p=(x0 - s, y))) == newi((x0 - s, y0), (x1, y)) and tuple(bi(15, b, p=(x1 + s, y))) == newi((x0, y0), (x1 + s, y))
    x = _rint(lerp(t, f32(x0), f32(x1)))
    assert tuple(bi(15, b, p=(x, y0 - s))) == newi((x0, y0 - s), (x1, y1)) and tuple(bi(15, b, p=(x, y1 + s))) == newi((x0, y0), (x1, y1 + s))
    yi = _rint(lerp(ty, f32(y0), f32(y1)))
    assert tuple(bi(15, b, p=(x, yi))) == b  # interior point


@replay(B, "union_non_empty_bounds2f_with_exterior_point_returns_non_empty_bounds2f", "union_non_empty_bounds2f_with_interior_point_returns_same_bounds2f")
@S
@given(v2s, fr(0.001, 10.0), fr(0.001, 10.0), fr(0.0, 1.0), fr(0.0, 1.0), fr(0.0, 1.0))
def test_b2f_union_point(p, dx, dy, s, t, ty):
    px, py, dx, dy, s = (f32(v) for v in (p[0], p[1], dx, dy, s))
    b = newf((px - dx, py - dy), (px + dx, py + dy))
    x0, y0, x1, y1 = (f32(v) for v in b)

    def union_is_componentwise(q):  # the definition (bounds2.rs:248-257); where lerp lands inside the box it is what the reference test spells out
        eq(bf(15, b, p=q), F(pmin(x0, q[0]), pmin(y0, q[1]), pmax(x1, q[0]), pmax(y1, q[1])))
    for tt in (f32(t) - f32(1.0), f32(t), f32(t) + f32(1.0)):
        y = lerp(tt, y0, y1)
        union_is_componentwise((x0 - s, y)); union_is_componentwise((x1 + s, y))
    x = lerp(t, x0, x1)
    union_is_componentwise((x, y0 - s)); union_is_componentwise((x, y1 + s))
    eq(bf(15, b, p=(x0 - s, lerp(t, y0, y1)))[[0, 2]], F(x0 - s, x1))
    xi, yi = lerp(t, x0, x1), lerp(ty, y0, y1)
    if x0 <= xi <= x1 and y0 <= yi <= y1:
        eq(bf(15, b, p=(xi, yi)), F(*b))


@replay(B, "union_non_empty_non_overlapping_bounds2f_returns_non_empty_bounds2f", "union_non_empty_overlapping_bounds2f_returns_non_empty_bounds2f",
        "intersect_non_empty_non_overlapping_bounds2f_returns_empty", "intersect_non_empty_overlapping_bounds2f_returns_non_empty")
@S
@given(v2s, fr(0.001, 10.0), fr(0.001, 10.0), fr(0.002, 10.0), fr(-2.0, 2.0), fr(-2.0, 2.0), fr(-2.0, 2.0), fr(-2.0, 2.0))
def test_b2f_union_and_intersect_of_boxes(p, dx, dy, s, t1, t2, s1, s2):
    px, py, dx, dy, s = (f32(v) for v in (p[0], p[1], dx, dy, s))
    b1 = newf((px - dx, py - dy), (px + dx, py + dy))
    x0, y0, x1, y1 = (f32(v) for v in b1)
    xa, xb, ya, yb = lerp(t1, x0, x1), lerp(t2, x0, x1), lerp(s1, y0, y1), lerp(s2, y0, y1)
    k = f32(0.001)
    disjoint = [newf((x0 - s, ya), (x0 - k, yb)), newf((x1 + k, ya), (x1 + s, yb)), newf((xa, y0 - s), (xb, y0 - k)), newf((xa, y1 + k), (xb, y1 + s))]
    for b2 in disjoint + [newf((xa, ya), (xb, yb))]:
        u = F(pmin(x0, f32(b2[0])), pmin(y0, f32(b2[1])), pmax(x1, f32(b2[2])), pmax(y1, f32(b2[3])))
        eq(bf(16, b1, b2), u); eq(bf(16, b2, b1), u)
        i = F(pmax(x0, f32(b2[0])), pmax(y0, f32(b2[1])), pmin(x1, f32(b2[2])), pmin(y1, f32(b2[3])))
        eq(bf(17, b1, b2), i); eq(bf(17, b2, b1), i)
    for j, b2 in enumerate(disjoint):
        # boxes placed strictly beside b1 have an empty intersection — wherever f32 keeps the 0.001 gap (|coordinates| < 100 + 20: one ulp is 8e-6)
        if (j < 2 and (x0 - k < x0) and (x1 + k > x1)) or (j >= 2 and (y0 - k < y0) and (y1 + k > y1)):
            assert bf(3, tuple(bf(17, b1, b2)), n=1)[0] == 1.0


@replay(B, "iterate_bounds2i_returns_grid_points_left_to_right_x_and_top_to_bottom_y", "iterate_bounds2i_with_0_in_one_dimension_returns_grid_points_along_the_other")
@S
@given(p2i, st.integers(1, 9), st.integers(1, 9))
def test_b2i_iteration(p, dx, dy):
    out = bi(18, newi(p, (p[0] + dx, p[1] + dy)))
    assert out[0] == dx * dy
    assert [tuple(out[1 + 2 * k: 3 + 2 * k]) for k in range(dx * dy)] == [(p[0] + x, p[1] + y) for y in range(dy) for x in range(dx)]
    out1, out2 = bi(18, newi(p, (p[0], p[1] + dy))), bi(18, newi(p, (p[0] + dx, p[1])))  # a zero-width box is walked as one column, a zero-height box as one row
    assert out1[0] == dy and [tuple(out1[1 + 2 * k: 3 + 2 * k]) for k in range(dy)] == [(p[0], p[1] + i) for i in range(dy)]
    assert out2[0] == dx and [tuple(out2[1 + 2 * k: 3 + 2 * k]) for k in range(dx)] == [(p[0] + i, p[1]) for i in range(dx)]


# ---------------------------------------------------------------------------------------------------------------- accounting
def not_applicable(file, name, should_panic):
    if should_panic:
        return "asserts a debug-assertion panic (zero divisor / zero-length normalize / index out of range / singular matrix): the restatement has no panics to pin"
    if name.endswith("_i32"):
        return "i32 instantiation of a generic vector type: the path only instantiates it at f32 (pixel coordinates are plain ints in the oracle; Bounds2i IS replayed)"
    if file.endswith("ray.rs") and name in ("has_nans", "scale_differentials_none"):
        return "Option<RayDifferential> / NaN bookkeeping of the Ray struct, no arithmetic"
    return None


def test_every_reference_test_is_accounted_for():
    total = replayed = 0
    unaccounted = []
    for file, tests in MANIFEST.items():
        for t in tests:
            total += 1
            if (file, t["name"]) in REPLAYS and not t["should_panic"]:
                assert REPLAYS[(file, t["name"])] in globals()
                replayed += 1
            elif not_applicable(file, t["name"], t["should_panic"]) is None:
                unaccounted.append((file, t["name"], t["line"]))
    assert total == 277
    assert not unaccounted, unaccounted
    assert replayed >= 180, replayed
    for key in REPLAYS:
        assert any(key[0] == f and any(t["name"] == key[1] for t in ts) for f, ts in MANIFEST.items()), key  # no replay of a test the reference does not hold
