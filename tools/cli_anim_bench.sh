#!/bin/bash
# tools/cli_anim_bench.sh [frames] -- end-to-end animation throughput of henjou_cli (C2 settings, N frames): serial output stage
# vs the overlapped one.  Run on the GPU box from the repo root.
N=${1:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$(mktemp -d)
cp -r $R/henjou-renderer_amd/assets/Model $W/Model
python3 - "$R" "$W" "$N" <<'PY'
import json, sys
r, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
ro = json.load(open(r + '/henjou-renderer_amd/assets/render_option_c2.json'))
ro['Animation'].update(start_frame=1, end_frame=1 + n)
json.dump(ro, open(w + '/render_option.json', 'w'))
PY
cd $W
for mode in 1 0; do
  python3 - "$mode" <<'PY'
import json, sys
ro = json.load(open('render_option.json'))
ro['Henjou_HIP'] = {'serial_io': sys.argv[1] == '1'}
json.dump(ro, open('render_option.json', 'w'))
PY
  echo "serial_io=$mode"
  timeout -k 10 300 $R/henjou-renderer_amd/henjou_cli render_option.json 2>&1 | grep -E "wall|frame 1:|error" 
  md5sum cornelbox_c2_00*.png | md5sum
done
rm -rf $W
