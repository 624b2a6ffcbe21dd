#!/bin/bash
# GPU box: the rocprofv3 runs whose summaries go under profiles/ (tools/profile_summarise.py copies them).
#   tools/profile_round.sh <tag>        e.g. r02_a
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python3 -c "import importlib,sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); print(importlib.import_module('ray-tracer_amd').build_id())" > $out/build_id.txt
cd /tmp && export TMPDIR=/tmp
# 1. the bench command under --kernel-trace --stats (per-kernel averages must agree with bench.py's HIP events).
#    --headline-only: the other legs (D2H loop, f64 report, other configurations) launch the same kernel instantiations
#    on other frames or next to copy kernels and would pool into the same rows; with them off every k_trace_shade
#    launch in the trace is one of the headline frame, running alone.
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --headline-only > $out/bench.log 2>&1
grep "^{\"metric\"" $out/bench.log > $out/bench.json
#    and the whole default command, unprofiled (what the driver runs)
timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py > $out/bench_full.log 2>&1
grep "^{\"metric\"" $out/bench_full.log > $out/bench_full.json
# 2. PMC passes (own runs, no tracing) over plain frames of the headline scene: HBM traffic, cache, issue counters
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
            "mem:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  RTX_PROFILE=0 RTX_LANES=1 timeout -s KILL 120 rocprofv3 --pmc $ctrs -d $out/pmc_composite4k_$name -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py composite:1 3840 2160 1 > $out/pmc_composite4k_$name.log 2>&1
done
# 3. kernel statistics of the other configurations (plain frames, one lane)
for spec in "teapot:teapot:1 3840 2160 10" "composite4k:composite:1 3840 2160 10" "composite8k:composite:1 7680 4320 5" \
            "glass_spheres:glass_spheres 1920 1080 10" "fresnel:fresnel 1920 1080 10" "cylinder_world:cylinder_world 3840 2160 5"; do
  name=${spec%%:*}; args=${spec#*:}
  RTX_PROFILE=0 RTX_LANES=1 timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $out/$name -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py $args > $out/$name.log 2>&1
done
echo done
