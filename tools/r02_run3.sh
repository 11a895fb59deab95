#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run3.txt
{
echo "== megakernel (refactored)"
timeout -k 10 120 $K $L render_option_c2.json --reps 3
timeout -k 10 120 $K $L render_option_c2.json --reps 3 --aovs
echo "== wavefront small first (64x64x4)"
HJR_PIPELINE=wf timeout -k 10 60 $K $L render_option_c2.json --width 64 --height 64 --spp 4 --reps 1
HJR_PIPELINE=mega timeout -k 10 60 $K $L render_option_c2.json --width 64 --height 64 --spp 4 --reps 1
} > $O 2>&1
cat $O
cd $R
HJR_T="tests/test_gpu_variants.py"
timeout -k 10 600 python -m pytest $HJR_T -m gpu -x -q -k "wavefront_default or wavefront_sizes" > gpurun_out/r02_pytest3.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r02_pytest3.log
cd $R/henjou-renderer_amd/assets
{
echo "== wavefront full size"
HJR_PIPELINE=wf timeout -k 10 120 $K $L render_option_c2.json --reps 3
HJR_PIPELINE=wf timeout -k 10 120 $K $L render_option_c2.json --reps 3 --aovs
HJR_PIPELINE=wf timeout -k 10 120 $K $L render_option_c2.json --reps 2 --integrator 2
HJR_PIPELINE=wf HJR_WF_CAP=1024 timeout -k 10 120 $K $L render_option_c2.json --reps 2
HJR_PIPELINE=wf HJR_WF_CAP=4096 HJR_LDS_STACK16=1 timeout -k 10 120 $K $L render_option_c2.json --reps 2
HJR_PIPELINE=wf HJR_LDS_STACK16=1 timeout -k 10 120 $K $L render_option_c2.json --reps 2
} >> $O 2>&1
tail -8 $O
