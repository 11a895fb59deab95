"""C++ multi-process test (CPU) of the tile partition / packing that henjou_cli's multi-GPU path and bench.py exchange over
RCCL: tests/native/tile_shard_test.cpp forks one process per rank, each packs its owned 8x8 tiles with the library's
hjr_pack_tiles, rank 0 reassembles with hjr_unpack_tiles and must get the frame back bit for bit."""
import os
import subprocess

import pytest

from scene_util import ROOT, hjr


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shard") / "tile_shard_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "native", "tile_shard_test.cpp"), "-o", out,
                           "-L" + hjr.PKG_DIR, "-lhenjou_hip", "-Wl,-rpath," + hjr.PKG_DIR, "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.mark.parametrize("w,h,n", [(64, 64, 2), (75, 41, 2), (1920, 1080, 8), (9, 17, 3), (8, 8, 4)])
def test_tile_shard_multi_process(exe, w, h, n):
    p = subprocess.run([exe, str(w), str(h), str(n)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "tile_shard_test ok" in p.stdout


def test_henjou_hip_section_devices_and_tile(tmp_path):
    """`Henjou_HIP.devices` selects the multi-GPU launcher of henjou_cli; `tile` other than 8 is rejected, not ignored."""
    import json
    base = json.load(open(os.path.join(hjr.ASSETS, "render_option_c1.json")))
    base["Henjou_HIP"] = {"devices": 8, "tile": 8}
    a = tmp_path / "a.json"
    a.write_text(json.dumps(base))
    o = hjr.load_render_option(str(a))
    assert o.devices == 8 and o.tile == 8
    assert hjr.load_render_option(os.path.join(hjr.ASSETS, "render_option_c1.json")).devices == 1
    for bad in ({"devices": 0}, {"devices": 2.5}, {"tile": 16}):
        base["Henjou_HIP"] = bad
        b = tmp_path / "b.json"
        b.write_text(json.dumps(base))
        with pytest.raises(hjr.HjrError):
            hjr.load_render_option(str(b))
