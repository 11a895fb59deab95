#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run8.txt
export HJR_PIPELINE=wf
{
timeout -k 5 30 $K $L render_option_c2.json --reps 3 &&
timeout -k 5 30 $K $L render_option_c2.json --reps 2 --aovs &&
HJR_WF_CAP=1024 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_WF_CAP=4096 HJR_LDS_STACK16=1 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_LDS_STACK16=1 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
for r in 4 8 24 32 48; do HJR_WF_REFILL=$r timeout -k 5 30 $K $L render_option_c2.json --reps 2 || break; done
for t in 1 16 64 128; do HJR_WF_TRACE_MIN=$t timeout -k 5 30 $K $L render_option_c2.json --reps 2 || break; done
timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 2
timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 1
echo "last rc $?"
} > $O 2>&1
cat $O
