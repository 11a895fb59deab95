#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so; W=$R/henjou-renderer_amd/build_v/libhenjou_hip_wd.so
O=$R/gpurun_out/r02_run12.txt
cd $R/henjou-renderer_amd/assets
export HJR_PIPELINE=wf
{
timeout -k 5 20 $K $W render_option_c2.json --reps 1 --aovs
echo "rc $?"
HJR_WF_CAP=1024 timeout -k 5 20 $K $W render_option_c2.json --reps 1
echo "rc $?"
} > $O 2>&1
cat $O
