#!/bin/bash
# first GPU pass of round 2: baseline numbers through kbench, the timing-diagnostic build, then the GPU test suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so; T=$R/henjou-renderer_amd/build_v/libhenjou_hip_timing.so
O=$R/gpurun_out/r02_run1.txt
{
timeout -k 10 120 $K $L render_option_c2.json --reps 5 --stats
timeout -k 10 120 $K $L render_option_c2.json --reps 5
timeout -k 10 120 $K $L render_option_c2.json --reps 3 --aovs
timeout -k 10 120 $K $L render_option_c2.json --reps 3 --integrator 2
timeout -k 10 120 $K $L render_option_c2.json --reps 3 --integrator 1
HJR_LDS_STACK16=1 timeout -k 10 120 $K $L render_option_c2.json --reps 3
HJR_LDS_BVH=0 timeout -k 10 120 $K $L render_option_c2.json --reps 3
timeout -k 10 120 $K $L render_option_c3.json --width 1920 --height 1080 --spp 256 --reps 2
timeout -k 10 120 $K $L render_option_c4.json --width 1920 --height 1080 --spp 256 --reps 2
timeout -k 10 120 $K $T render_option_c2.json --reps 1
} > $O 2>&1
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1
echo "pytest rc $?" >> $O
tail -3 gpurun_out/r02_pytest1.log >> $O
cat $O
