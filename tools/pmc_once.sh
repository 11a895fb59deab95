#!/bin/bash
# tools/pmc_once.sh <outname> "<counters>" [bench args]  -- one rocprofv3 --pmc pass over bench.py (no traces), prints per-dispatch
# values of the render kernel.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; CNT="$2"; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()): print(k, ' '.join('%.4g' % x for x in v))
PY
