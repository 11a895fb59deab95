#!/bin/bash
# Round-3 profile set, run on the GPU box from the repo root:  bash tools/prof_r03.sh
#  1. rocprofv3 --kernel-trace --stats of the bench command itself (python3 bench.py ...; counters off inside so that no child process runs under the tracer);
#  2. separate --pmc passes (never combined with traces) over tools/kbench for the launches the bench line quotes.
# Outputs under gpurun_out/prof_r03/; tools/summarize_prof_r03.py condenses them into profiles/r03_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc $?"
python3 $R/tools/make_stress_scene.py /tmp/stress --spheres 64 --segments 128 > /dev/null
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
cd $R/henjou-renderer_amd/assets
pass() { # name, counters, kbench args...
  local name=$1 counters=$2; shift; shift
  timeout -k 5 120 rocprofv3 --pmc $counters --output-format csv -d $OUT/$name -- $K $L "$@" --reps 1 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
for cfg in "nee_aovs render_option_c2.json --aovs" "nee_color render_option_c2.json" "nee_aovs_fast render_option_c2.json --aovs --fast" "mis render_option_c2.json --integrator 2" "stress /tmp/stress/render_option_stress.json"; do
  set -- $cfg; tag=$1; shift
  pass ${tag}_pmc_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@"
  pass ${tag}_pmc_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$@"
  pass ${tag}_pmc_cls "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM" "$@"
  pass ${tag}_pmc_fetch "FETCH_SIZE" "$@"
  pass ${tag}_pmc_write "WRITE_SIZE" "$@"
  pass ${tag}_pmc_tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "$@"
done
find $OUT -name "*.csv" | wc -l
