"""The file-level drop-in on a GPU: henjou_cli render_option.json == Renderer::initializeAndRender (renderer.h:1053-1317):
one <image_name>_<frame:03d>.png per frame in [start_frame, end_frame), equal to the oracle's frame pushed through the
reference's output stage (float4ConvertColor, bottom-up rows)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from scene_util import ROOT, Cornell, f32_time, hjr

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "henjou-renderer_amd", "henjou_cli")


def test_cli_renders_every_frame(tmp_path):
    assert os.path.exists(CLI), "henjou_cli is not built (python __graft_entry__.py)"
    work = tmp_path / "run"
    shutil.copytree(os.path.join(hjr.ASSETS, "Model"), work / "Model")
    ro = json.load(open(os.path.join(hjr.ASSETS, "render_option_c1.json")))
    ro["Image"].update(image_width=96, image_height=64, max_spp=20, image_name="clitest")
    ro["Animation"].update(start_frame=1, end_frame=3)
    ro["Henjou_HIP"] = {"seed": 9, "integrator": "NEE"}
    (work / "render_option.json").write_text(json.dumps(ro))
    (work / "fps.txt").write_text("24")
    p = subprocess.run([CLI, "render_option.json"], cwd=work, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    for frame in (1, 2):
        png = work / ("clitest_%03d.png" % frame)
        assert png.exists()
        got = hjr.load_png(str(png))
        assert got.shape == (64, 96, 4)
        # expected: oracle frame -> toSRGB/quantise -> rows bottom-up
        cwd = os.getcwd()
        os.chdir(work)
        try:
            opt = hjr.load_render_option("render_option.json")
            sc = hjr.Scene(opt.gltf_path.decode(), opt.gltf_name.decode(), opt)
        finally:
            os.chdir(cwd)
        t = f32_time(frame, opt.fps)
        cam = sc.camera(opt, t)
        arrays = sc.arrays(t)
        op = ob.make_params(96, 64, 20, cam.as_dict(), frame=frame, seed=9, sky=tuple(opt.scene_sky_default), ibl_intensity=opt.IBL_intensity)
        oc, _, _, _ = ob.OracleScene(arrays, ob.MATH_PORTABLE).render(op, want_aovs=False)
        exp = hjr.float4_to_srgb8(oc)[::-1]
        assert np.array_equal(got, exp), "frame %d: %d pixels differ" % (frame, int(np.sum(np.any(got != exp, axis=-1))))
    assert not (work / "clitest_003.png").exists()
    # error path: missing config -> non-zero exit, message on stderr
    q = subprocess.run([CLI, "nope.json"], cwd=work, capture_output=True, text=True, timeout=60)
    assert q.returncode != 0 and "not found" in q.stderr


def test_cli_rank_process_path_on_one_gpu(tmp_path):
    """The per-rank code of the multi-GPU launcher (packed tiles -> ncclGather -> hjr_unpack_tiles_device -> PNG) run as a world of
    one on the box's single GPU: the PNG must be byte-identical to the single-process path's.  (Two ranks need two GPUs: RCCL
    refuses two ranks on one device; the N > 1 collective itself is rehearsed over gloo in test_distributed_gloo.py and the tile
    partition by tests/native/tile_shard_test.cpp.)"""
    work = tmp_path / "run"
    shutil.copytree(os.path.join(hjr.ASSETS, "Model"), work / "Model")
    ro = json.load(open(os.path.join(hjr.ASSETS, "render_option_c1.json")))
    ro["Image"].update(image_width=75, image_height=41, max_spp=12, image_name="a")  # ragged frame: 10 x 6 tiles
    ro["Animation"].update(start_frame=1, end_frame=2)
    (work / "render_option.json").write_text(json.dumps(ro))
    p = subprocess.run([CLI, "render_option.json"], cwd=work, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    single = (work / "a_001.png").read_bytes()
    ro["Image"]["image_name"] = "b"
    (work / "render_option.json").write_text(json.dumps(ro))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    q = subprocess.run([CLI, "render_option.json", "--rank", "0", "--world", "1"], cwd=work, capture_output=True,
                       text=True, timeout=300, env=env)
    assert q.returncode == 0, q.stdout + q.stderr
    assert (work / "b_001.png").read_bytes() == single
    assert "render + gather + assemble" in q.stderr


def test_cli_launcher_stops_the_job_when_a_rank_fails(tmp_path):
    """--devices 2 on a box with ONE GPU: rank 1 cannot create its context (device ordinal out of range) and exits, rank 0 is set up and
    would wait for its peer inside ncclCommInitRank for ever.  The launcher must notice the failed child, stop the other and exit 1 —
    round 2's launcher waited for its children in order and hung with the GPU held."""
    import time
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    work = tmp_path / "run"
    shutil.copytree(os.path.join(hjr.ASSETS, "Model"), work / "Model")
    ro = json.load(open(os.path.join(hjr.ASSETS, "render_option_c1.json")))
    ro["Image"].update(image_width=64, image_height=64, max_spp=4, image_name="never")
    ro["Animation"].update(start_frame=1, end_frame=2)
    (work / "render_option.json").write_text(json.dumps(ro))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    p = subprocess.run([CLI, "render_option.json", "--devices", "2"], cwd=work, capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 1, (p.returncode, p.stderr[-1500:])
    assert time.time() - t0 < 60
    assert "stopping the others" in p.stderr and not (work / "never_001.png").exists()
