#!/bin/bash
# builds and runs scripts/sah_steps_check.cpp (CPU only); "quick" as the first argument runs the subset the test suite uses
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc -std=c++17 -O2 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -x hip -pthread $R/scripts/sah_steps_check.cpp $R/pbrt-v3-rs_amd/csrc/bvh_build.cpp -o $R/gpurun_out/sah_steps_check
$R/gpurun_out/sah_steps_check "$@"
