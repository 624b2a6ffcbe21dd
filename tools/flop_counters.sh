#!/bin/bash
# Hardware flop-instruction counters of the headline frame, f32 and f64 instantiations (GPU box):
#   tools/flop_counters.sh <tag>   -> gpurun_out/<tag>/flops_f32, flops_f64 (rocprofv3 --pmc, own runs, no tracing)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
RTX_PROFILE=0 RTX_LANES=1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
  -d $out/flops_f32 -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py composite:1 3840 2160 1 > $out/flops_f32.log 2>&1 || exit 1
RTX_F64=1 RTX_PROFILE=0 RTX_LANES=1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
  -d $out/flops_f64 -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/frame.py composite:1 3840 2160 1 > $out/flops_f64.log 2>&1 || exit 1
grep " rays " $out/flops_f32.log $out/flops_f64.log
