#!/bin/bash
# A/B of whole-frame time on a scene (GPU box): tools/ab_frame.sh "<frame.py args>" lib1[:ENV=V] lib2 ...
# Plain device-resident frames, one lane.
args=$1; shift
for spec in "$@"; do
  lib="${spec%%:*}"; envs=""
  if [[ "$spec" == *:* ]]; then envs="${spec#*:}"; fi
  out=$(env RTX_PROFILE=0 RTX_LANES=1 RTX_LIB=$PWD/ray-tracer_amd/$lib $envs timeout -k 10 120 python tools/frame.py $args 2>&1 | grep " rays ") || { echo "$spec FAILED: $out"; exit 1; }
  echo "$spec :: $(echo "$out" | sed 's/.*ms_total/ms_total/; s/ trace.*//')"
done
