#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/henjou-renderer_amd/assets
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so
O=$R/gpurun_out/r02_run7.txt
export TMPDIR=/tmp
{
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 3 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 2 --aovs &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 1 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 2 --integrator 2 &&
HJR_PIPELINE=wf HJR_WF_CAP=1024 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf HJR_LDS_STACK16=1 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf HJR_LDS_STACK16=1 HJR_WF_CAP=4096 timeout -k 5 30 $K $L render_option_c2.json --reps 2 &&
HJR_PIPELINE=wf timeout -k 5 30 $K $L render_option_c2.json --reps 1 --stats
echo "last rc $?"
for pipe in wf mega; do
  export HJR_PIPELINE=$pipe
  rm -rf /tmp/pmc_$pipe
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$pipe -- $K $L render_option_c2.json --reps 1 > /tmp/pmc_$pipe.log 2>&1
  echo "== pmc $pipe rc $?"
  python3 - /tmp/pmc_$pipe <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hjr_render' in r['Kernel_Name'] or 'hjr_wavefront' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()): print(k, ' '.join('%.4g' % x for x in v))
PY
  rm -rf /tmp/pmc2_$pipe
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc2_$pipe -- $K $L render_option_c2.json --reps 1 > /tmp/pmc2_$pipe.log 2>&1
  python3 - /tmp/pmc2_$pipe <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hjr_render' in r['Kernel_Name'] or 'hjr_wavefront' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()): print(k, ' '.join('%.4g' % x for x in v))
PY
done
} > $O 2>&1
cat $O
