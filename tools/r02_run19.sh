#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run19.txt
python3 $R/tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
python3 $R/tools/make_stress_scene.py /tmp/stress100k --spheres 16 --segments 128 > /dev/null
cd $R/henjou-renderer_amd/assets
{
for pipe in wf mega; do
export HJR_PIPELINE=$pipe
echo "== $pipe"
timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 || break
timeout -k 5 40 $K $L /tmp/stress/render_option_stress.json --reps 2 --aovs || break
timeout -k 5 60 $K $L /tmp/stress/render_option_stress.json --reps 1 --integrator 2 || break
timeout -k 5 40 $K $L /tmp/stress100k/render_option_stress.json --reps 2 || break
HJR_LDS_BVH=0 timeout -k 5 40 $K $L render_option_c2.json --reps 2 || break
done
echo "last rc $?"
} > $O 2>&1
cat $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/r02_pytest19.log 2>&1
echo "pytest variants rc $?"; tail -3 gpurun_out/r02_pytest19.log
