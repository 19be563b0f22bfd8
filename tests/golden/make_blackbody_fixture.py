"""RGB of the two `"blackbody L" [T scale]` light parameters the reference's scenes use (cameras/*.pbrt, objects/instances.pbrt: [3000 1.5]; lights/distant.pbrt: [4500 1.5])
-> blackbody_rgb.json.  Follows ParamSet::add_blackbody_spectrum (core/src/paramset/mod.rs:236-249): Planck's law normalised by its Wien maximum at the 471 CIE wavelengths
(spectrum/common.rs:361-398), RGBSpectrum::from(samples) (rgb_spectrum.rs:82-103: sum against the CIE matching curves, scale, XYZ -> RGB), times the scale — all in f32.
The matching curves are read from the reference's text (spectrum/cie.rs) at generation time; only the six resulting numbers are kept.  Blackbody / spectral parameters are
outside the product's scope: the fixture exists so that the reference's renders of those scenes can be used as expected outputs.
Run in the build container: python3 tests/golden/make_blackbody_fixture.py"""
import json
import os
import re

import numpy as np

f32 = np.float32
src = open("/root/reference/core/src/spectrum/cie.rs").read()


def table(name):
    m = re.search(r"pub const %s: \[Float; CIE_SAMPLES\] = \[(.*?)\];" % name, src, re.S)
    v = [f32(x) for x in re.findall(r"[-+0-9.eE]+", m.group(1))]
    assert len(v) == 471
    return v


X, Y, Z = table("CIE_X"), table("CIE_Y"), table("CIE_Z")
CIE_Y_INTEGRAL = f32(float(re.search(r"CIE_Y_INTEGRAL: Float = ([0-9.]+)", src).group(1)))


def planck(lam_nm, t):
    c, h, kb = f32(299792458.0), f32(6.62606957e-34), f32(1.3806488e-23)
    l = f32(lam_nm) * f32(1e-9)
    lambda5 = (l * l) * (l * l) * l
    return (f32(2.0) * h * c * c) / (lambda5 * (np.exp((h * c) / (l * kb * t), dtype=f32) - f32(1.0)))


def blackbody_rgb(t, scale):
    t = f32(t)
    with np.errstate(all="ignore"):
        lam_max = f32(2.8977721e-3) / t * f32(1e9)
        mx = planck(lam_max, t)
        xyz = [f32(0), f32(0), f32(0)]
        for i in range(471):
            val = planck(f32(360 + i), t) / mx
            xyz[0] += val * X[i]; xyz[1] += val * Y[i]; xyz[2] += val * Z[i]
    k = f32(830 - 360) / (CIE_Y_INTEGRAL * f32(471))
    xyz = [v * k for v in xyz]
    rgb = [f32(3.240479) * xyz[0] - f32(1.537150) * xyz[1] - f32(0.498535) * xyz[2],
           f32(-0.969256) * xyz[0] + f32(1.875991) * xyz[1] + f32(0.041556) * xyz[2],
           f32(0.055648) * xyz[0] - f32(0.204043) * xyz[1] + f32(1.057311) * xyz[2]]
    return [float(f32(scale) * v) for v in rgb]


out = {"3000x1.5": blackbody_rgb(3000, 1.5), "4500x1.5": blackbody_rgb(4500, 1.5)}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "blackbody_rgb.json"), "w"), indent=1)
print(out)
