#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run21.txt
python3 $R/tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
python3 $R/tools/make_stress_scene.py /tmp/stress100k --spheres 16 --segments 128 > /dev/null
cd $R/henjou-renderer_amd/assets
export HJR_PIPELINE=wf
{
for cap in 4096 8192 16384; do
echo "== cap $cap"
HJR_WF_CAP=$cap timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 || break
HJR_WF_CAP=$cap HJR_SHORT_STACK=8 timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 || break
HJR_WF_CAP=$cap timeout -k 5 40 $K $L /tmp/stress100k/render_option_stress.json --reps 2 || break
HJR_WF_CAP=$cap timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 1 --integrator 2 || break
done
echo "last rc $?"
} > $O 2>&1
cat $O
