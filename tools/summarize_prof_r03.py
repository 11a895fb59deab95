#!/usr/bin/env python3
"""Condenses gpurun_out/prof_r03 (tools/prof_r03.sh) into profiles/r03_kernel_stats.csv, profiles/r03_trace_bench.json and
profiles/r03_pmc.json (per configuration and kernel: mean counter values per dispatch, derived active-lane fraction, VALU
lane-operation rate and corrected HBM bytes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "prof_r03")
out = os.path.join(root, "profiles")
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, "r03_kernel_stats.csv"))
tb = os.path.join(src, "trace_bench.json")
if os.path.exists(tb):
    lines = [l for l in open(tb) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(out, "r03_trace_bench.json"), "w"), indent=1)
summary = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(src, "*_pmc_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d).split("_pmc_")[0]
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "hjr_" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            summary.setdefault(cfg, collections.OrderedDict()).setdefault(k, collections.OrderedDict())[c] = sum(v) / len(v)
for cfg, kernels in summary.items():
    for k, cs in kernels.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:  # KiB counters; gfx950 FETCH_SIZE reports half of wide reads (MI355X_MICROARCH.md, HBM)
            cs["hbm_bytes_per_launch_corrected"] = cs["FETCH_SIZE"] * 1024 * 2 + cs["WRITE_SIZE"] * 1024
        if "SQ_THREAD_CYCLES_VALU" in cs and cs.get("SQ_ACTIVE_INST_VALU"):
            cs["active_lane_frac"] = cs["SQ_THREAD_CYCLES_VALU"] / cs["SQ_ACTIVE_INST_VALU"] / 64.0
        if "GRBM_GUI_ACTIVE" in cs and cs.get("SQ_INSTS_VALU"):
            cs["simd_cycles_per_valu_inst"] = cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / cs["SQ_INSTS_VALU"]
json.dump(summary, open(os.path.join(out, "r03_pmc.json"), "w"), indent=1)
for cfg, kernels in summary.items():
    for k, cs in kernels.items():
        if "render_kernel" in k or "wavefront_kernel" in k:
            print(cfg, k[:70], {c: ("%.4g" % v) for c, v in cs.items() if c in ("SQ_INSTS_VALU", "active_lane_frac", "simd_cycles_per_valu_inst", "hbm_bytes_per_launch_corrected", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")})
