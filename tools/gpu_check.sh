#!/bin/bash
# GPU box: the -m gpu suite, then a few frame timings.  tools/gpu_check.sh <tag>
tag=${1:-check}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/gpu_tests.log 2>&1
rc=$?
tail -5 $out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for a in "teapot:1 3840 2160 10" "composite:1 3840 2160 10" "composite:1 7680 4320 5" "glass_spheres 1920 1080 10" "cylinder_world 3840 2160 5"; do
  RTX_PROFILE=0 timeout -k 10 120 python tools/frame.py $a 2>&1 | grep " rays " || exit 1
done
