#!/bin/bash
# One rocprofv3 --pmc pass over a frame.py run (GPU box).  tools/pmc_pass.sh <tag> "<counters>" <frame.py args...>
# Frames are plain device-resident frames (RTX_PROFILE=0, one lane): the kernels a frame loop runs.
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
RTX_PROFILE=0 RTX_LANES=1 timeout -s KILL 90 rocprofv3 --pmc $ctrs -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
