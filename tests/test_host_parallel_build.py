"""The per-frame host preparation (flatten, BVH build, node emit) runs on worker threads for large scenes; what it emits must
not depend on the number of threads (every parallel loop is element-wise or merges exact partial results)."""
import os
import re
import subprocess
import sys

from scene_util import ROOT


def test_frame_build_is_thread_count_invariant(tmp_path):
    exe = str(tmp_path / "frame_build_bench")
    host = os.path.join(ROOT, "henjou-renderer_amd", "host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + ROOT, os.path.join(ROOT, "tools", "frame_build_bench.cpp"),
                           os.path.join(host, "loaders.cpp"), os.path.join(host, "frame.cpp"), os.path.join(host, "image_io.cpp"), os.path.join(host, "jpeg.cpp"),
                           "-lz", "-pthread", "-o", exe])
    sdir = str(tmp_path / "scene")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_stress_scene.py"), sdir, "--spheres", "12", "--segments", "96"],
                          stdout=subprocess.DEVNULL)
    lines = {}
    for threads in ("1", "3", "8"):
        out = subprocess.run([exe, sdir, "render_option_stress.json", "1", threads], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        m = re.search(r"\((\d+) tris, (\d+) nodes, width (\d+), depth (\d+), stack (\d+), lds_mode (\d+)\)\s+hash ([0-9a-f]+)", out.stdout)
        assert m, out.stdout
        lines[threads] = m.groups()
    assert int(lines["1"][0]) >= 65536, "scene too small to reach the parallel builder"
    assert lines["1"] == lines["3"] == lines["8"], lines
