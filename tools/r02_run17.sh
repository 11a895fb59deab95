#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest17.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest17.log
timeout -k 10 400 python bench.py > gpurun_out/r02_bench17.json 2> gpurun_out/r02_bench17.err
echo "bench rc $?"; cat gpurun_out/r02_bench17.json | cut -c1-1500; tail -3 gpurun_out/r02_bench17.err
