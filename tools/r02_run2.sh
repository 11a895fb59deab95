#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest2.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r02_pytest2.log
timeout -k 10 600 python bench.py > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err
echo "bench rc $?"; cat gpurun_out/r02_bench2.json; tail -5 gpurun_out/r02_bench2.err
