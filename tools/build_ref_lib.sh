#!/bin/bash
# Build the device code of a git revision as ray-tracer_amd/libx_<tag>.so for A/B runs.
#   tools/build_ref_lib.sh <rev> <tag>
set -e
rev=$1; tag=$2
tmp=$(mktemp -d)
git -C /root/repo archive $rev ray-tracer_amd include | tar -x -C $tmp
make -s -C $tmp/ray-tracer_amd -j8 libraytracer_amd.so
cp $tmp/ray-tracer_amd/libraytracer_amd.so /root/repo/ray-tracer_amd/libx_$tag.so
rm -rf $tmp
