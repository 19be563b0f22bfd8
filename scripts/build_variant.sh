#!/bin/bash
# another build of the library next to the shipping one, for same-box A/B runs:  bash scripts/build_variant.sh NAME "-DPH_SOMETHING=0 ..."
# -> pbrt-v3-rs_amd/libpbrt_hip_NAME.so (git-ignored; travels with gpurun); use it with PBRT_HIP_LIB=$PWD/pbrt-v3-rs_amd/libpbrt_hip_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2
B=$R/gpurun_out/build_$NAME; mkdir -p $B
FLAGS="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -pthread $EXTRA"
cd $R/pbrt-v3-rs_amd/csrc
for f in api textures_api wavefront multi bvh_device bvh_sah_device; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $B/$f.o & done
for f in bvh_build host_setup; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f.cpp -o $B/$f.o & done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o $R/pbrt-v3-rs_amd/libpbrt_hip_$NAME.so $B/*.o -ldl
echo built $R/pbrt-v3-rs_amd/libpbrt_hip_$NAME.so
