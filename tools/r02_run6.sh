#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; W=$R/henjou-renderer_amd/build_v/libhenjou_hip_wd.so; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run6.txt
{
HJR_PIPELINE=wf timeout -k 5 30 $K $W render_option_c2.json --width 1920 --height 1080 --spp 32 --reps 1 &&
HJR_PIPELINE=mega timeout -k 5 30 $K $W render_option_c2.json --width 1920 --height 1080 --spp 32 --reps 1 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $W render_option_c2.json --reps 1 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 3 &&
HJR_PIPELINE=mega timeout -k 5 30 $K $L render_option_c2.json --reps 3 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 3 --aovs &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 1 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 2 &&
HJR_PIPELINE=wf HJR_WF_CAP=1024 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf HJR_LDS_STACK16=1 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf HJR_LDS_STACK16=1 HJR_WF_CAP=4096 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 1 --stats
echo "last rc $?"
} > $O 2>&1
cat $O
