#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; L=$R/henjou-renderer_amd/build_v/libhenjou_hip_wd.so
O=$R/gpurun_out/r02_run4.txt
export HJR_PIPELINE=wf
{
timeout -k 5 20 $K $L render_option_c2.json --width 256 --height 256 --spp 16 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --width 512 --height 512 --spp 32 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --width 1920 --height 1080 --spp 8 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --width 1920 --height 1080 --spp 16 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --width 1920 --height 1080 --spp 64 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --width 1920 --height 1080 --spp 64 --reps 1 --integrator 1 &&
timeout -k 5 30 $K $L render_option_c2.json --reps 1 --integrator 1 &&
timeout -k 5 30 $K $L render_option_c2.json --reps 1
echo "last rc $?"
} > $O 2>&1
cat $O
