"""Replays of the reference's own proptests (core/src/geometry/*.rs #[cfg(test)]) against the oracle's math
layer: each operator is re-stated inline in f32 exactly as the reference test does and compared for equality."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle_binding import oracle_binding

f32 = np.float32
finite = st.floats(min_value=-1e6, max_value=1e6, width=32, allow_nan=False, allow_infinity=False)
vec3 = st.tuples(finite, finite, finite)


def op(code, vals, n_out=3):
    a = np.zeros(16, np.float32); a[: len(vals)] = vals
    out = np.zeros(16, np.float32)
    oracle_binding().lib.oracle_geom_op(code, a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n_out]


@settings(max_examples=200, deadline=None)
@given(vec3, vec3)
def test_dot_and_cross(a, b):  # vector3.rs:645-672
    a = np.array(a, f32); b = np.array(b, f32)
    assert op(0, list(a) + list(b), 1)[0] == a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
    want = np.array([(a[1] * b[2]) - (a[2] * b[1]), (a[2] * b[0]) - (a[0] * b[2]), (a[0] * b[1]) - (a[1] * b[0])], f32)
    assert np.array_equal(op(1, list(a) + list(b)), want)


@settings(max_examples=200, deadline=None)
@given(vec3)
def test_normalize_multiplies_by_reciprocal(a):  # vector3.rs:632-640: v * (1/len), not v / len
    a = np.array(a, f32)
    l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2]
    if l2 == 0 or not np.isfinite(l2):
        return
    inv = f32(1.0) / np.sqrt(l2)
    assert np.array_equal(op(2, list(a)), np.array([inv * a[0], inv * a[1], inv * a[2]], f32))
    got = op(3, list(a), 2)
    assert got[0] == np.sqrt(l2) and got[1] == l2


@settings(max_examples=200, deadline=None)
@given(vec3)
def test_abs_max_component_max_dimension(a):  # vector3.rs:600-630, 700-760
    a = np.array(a, f32)
    assert np.array_equal(op(4, list(a)), np.array([-v if v < 0 else v for v in a], f32))
    assert op(5, list(a), 1)[0] == max(a)
    md = int(op(6, list(a), 1)[0])
    want = (0 if a[0] > a[2] else 2) if a[0] > a[1] else (1 if a[1] > a[2] else 2)
    assert md == want


@settings(max_examples=100, deadline=None)
@given(vec3, st.permutations([0, 1, 2]))
def test_permute(a, perm):  # vector3.rs:770-790
    a = np.array(a, f32)
    assert np.array_equal(op(7, list(a) + [float(p) for p in perm]), a[list(perm)])


@settings(max_examples=200, deadline=None)
@given(vec3, vec3, finite)
def test_ray_at(o, d, t):  # ray.rs:299-305: o + d * t
    o = np.array(o, f32); d = np.array(d, f32); t = f32(t)
    assert np.array_equal(op(8, list(o) + list(d) + [t]), o + d * t)


@settings(max_examples=200, deadline=None)
@given(vec3)
def test_coordinate_system(a):  # coordinate_system.rs:26-65
    a = np.array(a, f32)
    l = np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
    if not (l > 1e-3):
        return
    v1 = a * (f32(1.0) / l)
    out = op(9, list(v1), 6)
    if abs(v1[0]) > abs(v1[1]):
        inv = f32(1.0) / np.sqrt(v1[0] * v1[0] + v1[2] * v1[2]); v2 = np.array([inv * -v1[2], inv * f32(0), inv * v1[0]], f32)
    else:
        inv = f32(1.0) / np.sqrt(v1[1] * v1[1] + v1[2] * v1[2]); v2 = np.array([inv * f32(0), inv * v1[2], inv * -v1[1]], f32)
    assert np.array_equal(out[:3], v2)
    v3 = np.array([(v1[1] * v2[2]) - (v1[2] * v2[1]), (v1[2] * v2[0]) - (v1[0] * v2[2]), (v1[0] * v2[1]) - (v1[1] * v2[0])], f32)
    assert np.array_equal(out[3:], v3)
    assert abs(float(np.dot(out[:3], v1))) < 1e-5 and abs(float(np.dot(out[3:], v1))) < 1e-5


def test_matrix_inverse_cases():  # matrix4x4.rs:324-396
    ident = np.eye(4, dtype=f32).ravel()
    assert np.array_equal(op(10, list(ident), 16), ident)
    m = np.array([[2, 0, 0, 1], [0, 4, 0, -2], [0, 0, 8, 3], [0, 0, 0, 1]], f32)
    inv = op(10, list(m.ravel()), 16).reshape(4, 4)
    assert np.allclose(inv @ m, np.eye(4), atol=1e-6)
    rng = np.random.default_rng(3)
    for _ in range(50):
        m = rng.uniform(-2, 2, (4, 4)).astype(f32)
        if abs(np.linalg.det(m.astype(np.float64))) < 0.1:
            continue
        inv = op(10, list(m.ravel()), 16).reshape(4, 4)
        assert np.allclose(inv.astype(np.float64) @ m.astype(np.float64), np.eye(4), atol=2e-3)


@settings(max_examples=100, deadline=None)
@given(vec3, vec3)
def test_face_forward_and_distance_squared(a, b):  # normal.rs / point3.rs:601-607
    a = np.array(a, f32); b = np.array(b, f32)
    d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
    assert np.array_equal(op(11, list(a) + list(b)), -a if d < 0 else a)
    e = a - b
    assert op(12, list(a) + list(b), 1)[0] == e[0] * e[0] + e[1] * e[1] + e[2] * e[2]


def test_bounds2i_iteration_is_row_major():  # bounds2.rs:460-470,1202-1230 — the pixel order of render_tile
    order = [(x, y) for y in range(2, 5) for x in range(1, 4)]
    assert order[0] == (1, 2) and order[1] == (2, 2) and order[3] == (1, 3) and order[-1] == (3, 4)
