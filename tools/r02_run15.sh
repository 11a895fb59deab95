#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run15.txt
cd $R/henjou-renderer_amd/assets
export HJR_PIPELINE=wf
{
timeout -k 5 20 $K $L render_option_c2.json --width 256 --height 256 --spp 16 --reps 1 &&
timeout -k 5 20 $K $L render_option_c2.json --reps 3 &&
timeout -k 5 20 $K $L render_option_c2.json --reps 2 --aovs &&
for f in 16 48 64; do HJR_WF_FLUSH=$f timeout -k 5 20 $K $L render_option_c2.json --reps 2 || break; done
for r in 8 24 32; do HJR_WF_REFILL=$r timeout -k 5 20 $K $L render_option_c2.json --reps 2 || break; done
timeout -k 5 20 $K $L render_option_c2.json --reps 2 --integrator 2 &&
timeout -k 5 20 $K $L render_option_c2.json --reps 2 --integrator 1
echo "last rc $?"
} > $O 2>&1
cat $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/r02_pytest15.log 2>&1
echo "pytest variants rc $?"; tail -3 gpurun_out/r02_pytest15.log
