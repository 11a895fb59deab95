#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/r02_pytest22.log 2>&1
echo "pytest rc $?"; tail -6 gpurun_out/r02_pytest22.log
