#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run20.txt
python3 $R/tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
cd $R/henjou-renderer_amd/assets
{
for lm in 1 2 3 4; do
echo "== leaf max $lm"
HJR_LEAF_MAX=$lm HJR_PIPELINE=mega timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 --stats || break
HJR_LEAF_MAX=$lm HJR_PIPELINE=wf timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 || break
done
echo "== wf knobs"
HJR_PIPELINE=wf HJR_SHORT_STACK=8 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
HJR_PIPELINE=wf HJR_SHORT_STACK=24 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
HJR_PIPELINE=wf HJR_WF_REFILL=32 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
HJR_PIPELINE=wf HJR_WF_REFILL=40 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
HJR_PIPELINE=wf HJR_WF_CAP=4096 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2
echo "last rc $?"
} > $O 2>&1
cat $O
