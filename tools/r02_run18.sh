#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so; S=$R/henjou-renderer_amd/build_v/libhenjou_hip_spec.so
O=$R/gpurun_out/r02_run18.txt
python3 $R/tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
cd $R/henjou-renderer_amd/assets
export HJR_PIPELINE=wf
{
timeout -k 5 20 $K $S render_option_c2.json --width 256 --height 256 --spp 16 --reps 1 &&
timeout -k 5 20 $K $S render_option_c2.json --reps 3 &&
timeout -k 5 20 $K $L render_option_c2.json --reps 3 &&
timeout -k 5 20 $K $S render_option_c2.json --reps 2 --integrator 2 &&
timeout -k 5 20 $K $S render_option_c2.json --reps 2 --aovs &&
HJR_WF_REFILL=24 timeout -k 5 20 $K $S render_option_c2.json --reps 2 &&
HJR_WF_REFILL=8 timeout -k 5 20 $K $S render_option_c2.json --reps 2 &&
timeout -k 5 40 $K $S /tmp/stress/render_option_stress.json --reps 2 &&
timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
echo "last rc $?"
} > $O 2>&1
cat $O
