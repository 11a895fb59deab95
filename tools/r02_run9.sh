#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run9.txt
cd $R
python tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
cd $R/henjou-renderer_amd/assets
{
echo "== stress 1M tris: mega vs wf"
HJR_PIPELINE=mega timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2 &&
HJR_PIPELINE=wf timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2 &&
HJR_PIPELINE=wf HJR_WF_REFILL=32 timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2 &&
HJR_PIPELINE=wf HJR_WF_REFILL=8 timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2 &&
HJR_PIPELINE=wf HJR_BVH_WIDTH=2 timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2 &&
HJR_PIPELINE=wf HJR_WF_CAP=4096 timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 2
echo "last rc $?"
} > $O 2>&1
cat $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/r02_pytest9.log 2>&1
echo "pytest variants rc $?"; tail -5 gpurun_out/r02_pytest9.log
HJR_PIPELINE=wf timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench.py > gpurun_out/r02_pytest9b.log 2>&1
echo "pytest all-with-wf rc $?"; tail -5 gpurun_out/r02_pytest9b.log
