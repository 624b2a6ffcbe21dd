#!/bin/bash
# A/B experiment driver (GPU box): runs bench.py --no-cpu once per library variant given.
#   tools/ab.sh libx_a.so[:ENV=V] libx_b.so ...
for spec in "$@"; do
  lib="${spec%%:*}"; envs=""
  if [[ "$spec" == *:* ]]; then envs="${spec#*:}"; fi
  out=$(env RTX_LIB=$PWD/ray-tracer_amd/$lib $envs timeout -k 10 120 python bench.py --no-cpu --steps 30 --warmup 5 2>&1 | tail -1) || { echo "$spec FAILED: $out"; exit 1; }
  echo "$spec $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step %.4f  Mrays/s %.0f  trace_avg_us %.1f frac %.3f" % (d["ms_per_step"], d["value"], 1000*d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))')"
done
