#!/bin/bash
# GPU box: the rocprofv3 runs whose summaries go under profiles/ (tools/profile_summarise.py copies them).
#   tools/profile_round.sh <tag>        e.g. r01_d
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. the bench command itself under --kernel-trace --stats (per-kernel averages must agree with bench.py's HIP events)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py > $out/bench.log 2>&1
grep "^{\"metric\"" $out/bench.log > $out/bench.json
# 2. PMC passes (own runs, no tracing) over one frame of each bench scene: HBM traffic and issue counters
for scene in glass_spheres fresnel; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 90 rocprofv3 --pmc $ctrs -d $out/pmc_${scene}_$name -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py $scene 1920 1080 1 > $out/pmc_${scene}_$name.log 2>&1
  done
done
# 3. the mesh scenes
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/teapot -o teapot --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py teapot:1 3840 2160 10 > $out/teapot.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/composite4k -o composite --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py composite:1 3840 2160 10 > $out/composite4k.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/composite8k -o composite --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py composite:1 7680 4320 5 > $out/composite8k.log 2>&1
echo done
