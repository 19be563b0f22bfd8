#!/usr/bin/env python3
"""Builds tests/golden/sobol_subset.npz from the reference's Sobol generator-matrix DATA tables
(core/src/sobol_matrices.rs: SOBOL_MATRICES_32, VD_C_SOBOL_MATRICES, VD_C_SOBOL_MATRICES_INV), read as text.

Only a subset is stored: the first N_DIMS dimensions (52 u32 each) and the first N_M VdC matrices, enough for the parity
tests (max_depth <= 4 at resolutions <= 256).  The full tables are supplied at run time by the host through
pbrt_hip_set_sobol_tables; the library embeds none.  Run here (the reference tree exists only in the build container):

    python tests/golden/make_sobol_fixture.py
"""
import os
import re

import numpy as np

REF = "/root/reference/core/src/sobol_matrices.rs"
N_DIMS, N_M = 48, 9


def table(src, name):
    m = re.search(r"pub const %s:[^=]*=\s*\[(.*?)\n\];" % name, src, re.S)
    assert m, name
    return [int(v, 16) if v.lower().startswith("0x") else int(v) for v in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", re.sub(r"//.*", "", m.group(1)))]


def main():
    src = open(REF).read()
    m32 = np.array(table(src, "SOBOL_MATRICES_32"), dtype=np.uint32)
    vdc = np.array(table(src, "VD_C_SOBOL_MATRICES"), dtype=np.uint64)
    vdci = np.array(table(src, "VD_C_SOBOL_MATRICES_INV"), dtype=np.uint64)
    assert m32.size == 1024 * 52 and vdc.size == 25 * 52 and vdci.size == 26 * 52, (m32.size, vdc.size, vdci.size)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sobol_subset.npz")
    np.savez_compressed(out, m32=m32[: N_DIMS * 52], vdc=vdc[: N_M * 52], vdc_inv=vdci[: N_M * 52])
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
