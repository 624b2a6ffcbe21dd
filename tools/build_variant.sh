#!/bin/bash
# Build an A/B variant of the library (CPU box): tools/build_variant.sh <name> <extra hipcc flags...>
#   -> ray-tracer_amd/libx_<name>.so   (RTX_LIB=$PWD/ray-tracer_amd/libx_<name>.so selects it; tools/ab_frame.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../ray-tracer_amd"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 -fno-slp-vectorize -DRT_BUILD"
/opt/rocm/bin/hipcc $F "$@" -c csrc/rtx_device.hip -o /tmp/v_${name}_device.o
/opt/rocm/bin/hipcc $F "$@" -c csrc/rtx_lbvh.hip -o /tmp/v_${name}_lbvh.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libx_${name}.so host/rt_host.o host/rt_libm.o host/rt_samples.o host/rt_obj.o /tmp/v_${name}_device.o csrc/rtx_multi.o /tmp/v_${name}_lbvh.o
echo built libx_${name}.so
