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


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 8), (3840, 2160, 8), (1920, 1080, 4), (1920, 1080, 2), (75, 41, 3)])
def test_tile_ids_partition_the_frame_and_run_along_diagonals(w, h, n):
    """Tile ids rotate every row of tiles by its row number (csrc/hjr_layout.h): still a partition with the round-robin tile counts
    of hjr_owned_tiles, the library's pack / unpack agree with the Python mask, and when tiles_x is a multiple of the GPU count (1080p
    and 4K on 2 / 4 / 8 GPUs) a rank no longer owns fixed vertical stripes: every column of tiles meets every rank."""
    import numpy as np
    masks = [hjr.owned_tile_mask(w, h, r, n) for r in range(n)]
    assert np.array_equal(np.sum(masks, axis=0), np.ones((h, w), dtype=np.int64))
    tiles = [m[::8, ::8] for m in masks]
    for r in range(n):
        assert int(tiles[r].sum()) == hjr.lib().hjr_owned_tiles(w, h, r, n)
    owner = np.zeros(tiles[0].shape, dtype=np.int64)
    for r in range(n):
        owner[tiles[r]] = r
    if owner.shape[0] >= n:
        for col in range(owner.shape[1]):
            assert len(set(owner[:, col].tolist())) == n, "tile column %d belongs to a subset of the ranks" % col
    # the library's own numbering: pack rank 1's tiles of a frame that stores (x, y) in every pixel and look at what arrived
    frame = np.zeros((h, w, 4), dtype=np.float32)
    frame[..., 0] = np.arange(w, dtype=np.float32)[None, :]
    frame[..., 1] = np.arange(h, dtype=np.float32)[:, None]
    frame[..., 3] = 1.0
    r = 1 % n
    packed = hjr.pack_tiles(frame, r, n)
    inside = packed[..., 3] == 1.0
    xs, ys = packed[..., 0][inside].astype(np.int64), packed[..., 1][inside].astype(np.int64)
    assert masks[r][ys, xs].all() and inside.sum() == masks[r].sum()


def test_cli_launcher_propagates_rank_failures_without_a_gpu(tmp_path):
    """The multi-GPU launcher of henjou_cli (fork + exec of one rank process per GPU, RCCL id over inherited pipes): when the ranks fail
    before they meet — here: the config does not exist, so no rank gets as far as a GPU call — the launcher reaps them, reports and exits 1
    instead of waiting for ever.  Host-only: runs on the CPU box."""
    cli = os.path.join(hjr.PKG_DIR, "henjou_cli")
    assert os.path.exists(cli), "henjou_cli is not built (python __graft_entry__.py)"
    p = subprocess.run([cli, str(tmp_path / "missing.json"), "--devices", "3"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1, (p.returncode, p.stderr[-800:])
    assert "not found" in p.stderr and "stopping the others" in p.stderr
    # a rank process started by hand without its pipe is refused, not run
    q = subprocess.run([cli, str(tmp_path / "missing.json"), "--rank", "1", "--world", "2"], capture_output=True, text=True, timeout=60)
    assert q.returncode == 2
