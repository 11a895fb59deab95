"""bench.py contract on a GPU: one JSON line with the required keys at N = 1, and a rehearsal of the N = 2 code path
(two ranks pinned to the one GPU of the box, gloo instead of RCCL) — the driver's N = 2/4/8 runs use the same code with nccl."""
import json
import os
import subprocess
import sys

import pytest

from scene_util import ROOT

pytestmark = pytest.mark.gpu
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"]


def run(cmd, env=None):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    d = run([sys.executable, "bench.py", "--width", "320", "--height", "200", "--spp", "32", "--steps", "2", "--warmup", "1", "--cpu-spp", "2",
             "--stress-spheres", "8", "--stress-segments", "32"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Msamples/s" and d["dtype"] == "f32"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    # the bound is chosen from counters collected inside the run: VALU lane-operations for a scene served by LDS / L2
    assert r["counters"]["source"].startswith("rocprofv3"), r["counters"]
    assert (r["bound"], r["peak"], r["unit"]) in (("valu", 157.3, "TFLOP/s"), ("hbm", 8000.0, "GB/s"))
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] > 0 and 0 < r["counters"]["active_lane_frac"] <= 1.0
    assert d["color_only"]["value"] > 0
    big = d["secondary"]["large_scene"]  # the memory-path scene (here a small one) with its measured fabric traffic
    assert "error" not in big, big
    assert big["value"] > 0 and big["triangles"] == 12 + 8 * 960 and big["roofline"]["bound"] == "hbm" and big["roofline"]["traffic"] > 0
    assert 0 < big["roofline"]["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Msamples/s"
    assert c["rmse_gpu_vs_cpu"]["value"] < 1e-3 * 50  # 2 spp here: a smoke value; the 1024-spp bar is checked in test_gpu_parity.py


def test_bench_two_rank_rehearsal():
    env = dict(os.environ, HJR_BENCH_DEVICE="0", HJR_BENCH_BACKEND="gloo")
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29577", "bench.py", "--gpus", "2", "--width", "320", "--height", "200", "--spp", "32", "--steps", "1",
             "--warmup", "1"], env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and "cpu_baseline" not in d
    assert "gather" in d["config"]["parallelism"]


def test_bench_gpus_2_started_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: bench.py starts torch.distributed.run itself as a child process and relays rank 0's
    line (rehearsal knobs: both ranks on GPU 0, gloo in place of RCCL — two RCCL ranks need two GPUs)."""
    env = dict(os.environ, HJR_BENCH_DEVICE="0", HJR_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    d = run([sys.executable, "bench.py", "--gpus", "2", "--width", "320", "--height", "200", "--spp", "32", "--steps", "1", "--warmup", "1"], env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "gather" in d["config"]["parallelism"]
