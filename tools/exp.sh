#!/bin/bash
# tools/exp.sh "<name>=<env assignments>" ...   e.g.  tools/exp.sh "base=" "w4=HJR_LIB=henjou-renderer_amd/build_v/libhenjou_hip_w4.so"
# Runs bench.py (kernel timing only) per variant and prints kernel ms / Msamples/s.
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%=*}"; envs="${spec#*=}"
  out=$(env $envs timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>&1 | grep '^{')
  echo "$name: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kernel_ms", r["kernel_ms_avg"], "Msps", d["value"], "frac", r["frac"], "box/samp", r["per_sample"]["box_tests_closest"]+r["per_sample"]["box_tests_shadow"], "tri/samp", r["per_sample"]["tri_tests_closest"]+r["per_sample"]["tri_tests_shadow"])' 2>&1)" | tee -a gpurun_out/exp.log
done
