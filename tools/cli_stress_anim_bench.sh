#!/bin/bash
# tools/cli_stress_anim_bench.sh [frames] -- henjou_cli on the 1 M-triangle scene with a forced per-frame rebuild: serial loop vs
# next-frame scene preparation overlapped with the render.  Run on the GPU box from the repo root.
N=${1:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$(mktemp -d)
python3 $R/tools/make_stress_scene.py $W > /dev/null
python3 - "$W" "$N" <<'PY'
import json, sys
w, n = sys.argv[1], int(sys.argv[2])
ro = json.load(open(w + '/render_option_stress.json'))
ro['Animation'].update(start_frame=1, end_frame=1 + n)
ro['Image'].update(image_width=1920, image_height=1080, max_spp=64)
json.dump(ro, open(w + '/render_option.json', 'w'))
PY
cd $W
for mode in 1 0; do
  python3 - "$mode" <<'PY'
import json, sys
ro = json.load(open('render_option.json'))
ro['Henjou_HIP'] = {'serial_io': sys.argv[1] == '1', 'force_rebuild': True}
json.dump(ro, open('render_option.json', 'w'))
PY
  echo "serial_io=$mode (forced rebuild every frame)"
  timeout -k 10 300 $R/henjou-renderer_amd/henjou_cli render_option.json 2>&1 | grep -E "wall|frame 2:|error"
done
rm -rf $W
