#!/bin/bash
# GPU box: kernel trace of a few frames of one scene, then the last frame's launches in order.
#   tools/trace_frame.sh <tag> <frame.py args...>     (RTX_LANES / RTX_PROFILE pass through)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
RTX_PROFILE=0 timeout -k 10 180 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py "$@" > $out/frame.log 2>&1 || { tail -5 $out/frame.log; exit 1; }
tail -1 $out/frame.log
python3 $GRAFT_REPO_ROOT/tools/trace_last_frame.py $out/t_kernel_trace.csv
