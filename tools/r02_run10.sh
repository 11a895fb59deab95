#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/tools/kbench; L=$R/henjou-renderer_amd/libhenjou_hip.so; T=$R/henjou-renderer_amd/build_v/libhenjou_hip_wft.so
O=$R/gpurun_out/r02_run10.txt
cd $R/henjou-renderer_amd/assets
export HJR_PIPELINE=wf TMPDIR=/tmp
{
timeout -k 5 30 $K $T render_option_c2.json --reps 1
HJR_WF_REFILL=32 timeout -k 5 30 $K $T render_option_c2.json --reps 1
timeout -k 5 30 $K $T render_option_c2.json --reps 1 --aovs
timeout -k 5 30 $K $T render_option_c2.json --reps 1 --integrator 2
rm -rf /tmp/pmc_wf
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_wf -- $K $L render_option_c2.json --reps 1 > /tmp/pmc_wf.log 2>&1
python3 - /tmp/pmc_wf <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hjr_render' in r['Kernel_Name'] or 'hjr_wavefront' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()): print(k, ' '.join('%.4g' % x for x in v))
PY
} > $O 2>&1
cat $O
