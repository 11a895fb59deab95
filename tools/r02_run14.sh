#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; T=$R/henjou-renderer_amd/build_v/libhenjou_hip_wft.so
cd $R/henjou-renderer_amd/assets
HJR_PIPELINE=wf timeout -k 5 20 $K $T render_option_c2.json --reps 1 2>&1 | tail -4 > $R/gpurun_out/r02_run14.txt
cat $R/gpurun_out/r02_run14.txt
