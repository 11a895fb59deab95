"""CPU sanitizer run (ASan + UBSan) of the product's host code and of the oracle.  GPU sanitizers are not available on the
pool, so this is where out-of-bounds / UB bugs of the host side get caught."""
import os
import subprocess

import pytest

from scene_util import ROOT, hjr


@pytest.mark.parametrize("config", ["render_option_c1.json", "render_option_tex.json"])
def test_host_and_oracle_under_asan_ubsan(tmp_path, config):
    exe = str(tmp_path / "sanitize_driver")
    host = os.path.join(ROOT, "henjou-renderer_amd", "host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
           os.path.join(ROOT, "tests", "native", "sanitize_driver.cpp"), os.path.join(host, "loaders.cpp"),
           os.path.join(host, "frame.cpp"), os.path.join(host, "image_io.cpp"), os.path.join(host, "jpeg.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "hjr_oracle.c"),
           "-o", exe, "-lz", "-lm", "-lpthread"]
    if not os.path.exists(exe):
        subprocess.check_call(cmd)
    import numpy as np
    from PIL import Image
    yy, xx = np.mgrid[0:40, 0:56]
    Image.fromarray(np.stack([(xx * 4) % 256, (yy * 6) % 256, (xx * yy) % 256], -1).astype(np.uint8)).save(str(tmp_path / "fuzz.jpg"), quality=88, subsampling=2)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, hjr.ASSETS, config, str(tmp_path)], capture_output=True, text=True, env=env, timeout=300, cwd=hjr.ASSETS)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    assert "sanitize_driver ok" in p.stdout
