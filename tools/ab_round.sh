#!/bin/bash
# GPU box: frame-time A/B of library variants over the usual frames.  tools/ab_round.sh lib1[:ENV=V] lib2 ...
cd $GRAFT_REPO_ROOT
for a in "composite:1 3840 2160 20" "teapot:1 3840 2160 20" "glass_spheres 1920 1080 20" "cylinder_world 3840 2160 10" "composite:1 7680 4320 5"; do
  echo "== $a"
  bash tools/ab_frame.sh "$a" "$@" || exit 1
done
