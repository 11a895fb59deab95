"""Denoise-mode replacement: oracle-side properties on the CPU, GPU == oracle bit for bit on the GPU.
The OptiX AI denoiser of the reference is a closed network: what is pinned here is the replacement's own specification
(henjou-renderer_amd/csrc/hjr_denoise.hip.h), restated independently in oracle/hjr_oracle.c."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from scene_util import ROOT, Cornell, f32_time, hjr


def _noisy_frame(w=72, h=40, spp=2):
    s = Cornell()
    oc, oa, on, _ = ob.OracleScene(s.arrays, ob.MATH_PORTABLE).render(s.oracle_params(w, h, spp))
    return s, oc, oa, on


def test_oracle_denoise_properties():
    rng = np.random.default_rng(5)
    h, w = 24, 40
    flat = np.full((h, w, 4), 0.37, np.float32)
    guide = np.zeros((h, w, 4), np.float32)
    # a constant image is a fixed point up to the rounding of sum / cum
    out = ob.denoise(1, flat, guide, guide)
    assert np.allclose(out, flat, rtol=2e-6, atol=0)
    # Default mode copies; alpha is carried through the filter
    img = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    assert np.array_equal(ob.denoise(0, img, guide, guide), img)
    assert np.array_equal(ob.denoise(1, img, guide, guide)[..., 3], img[..., 3])
    # noise on a flat guide is reduced; a guide edge is kept (two albedo regions with different means)
    base = np.zeros((h, w, 4), np.float32)
    base[:, : w // 2, :3] = 0.2
    base[:, w // 2:, :3] = 0.8
    noisy = base.copy()
    noisy[..., :3] += rng.normal(0, 0.05, (h, w, 3)).astype(np.float32)
    albedo = base.copy()
    out = ob.denoise(1, noisy, albedo, guide)
    assert np.std(out[:, : w // 2 - 1, 0]) < 0.5 * np.std(noisy[:, : w // 2 - 1, 0])
    assert abs(out[:, : w // 2 - 1, 0].mean() - 0.2) < 0.02 and abs(out[:, w // 2 + 1:, 0].mean() - 0.8) < 0.02
    # 2x upscale: output size, and a constant stays (bilinear weights sum to one exactly: 0.75 + 0.25)
    up = ob.denoise(2, flat, guide, guide)
    assert up.shape == (2 * h, 2 * w, 4) and np.allclose(up, 0.37, rtol=3e-6, atol=0)


def test_oracle_denoise_on_a_rendered_frame():
    """Against a 256 spp frame of another sample set: the 4 spp frame gets much closer once filtered (light sources and the
    constant background excluded)."""
    s = Cornell()
    o = ob.OracleScene(s.arrays, ob.MATH_PORTABLE)
    oc, oa, on, _ = o.render(s.oracle_params(96, 54, 4))
    ref, _, _, _ = o.render(s.oracle_params(96, 54, 256, frame=7))
    out = ob.denoise(1, oc, oa, on)
    assert np.isfinite(out).all()
    m = (ref[..., :3].max(axis=-1) < 3.0) & (np.abs(ref[..., :3] - 0.8).max(axis=-1) > 1e-3)

    def rmse(a):
        return float(np.sqrt(np.mean((a[..., :3][m] - ref[..., :3][m]) ** 2)))
    assert rmse(out) < 0.35 * rmse(oc), (rmse(oc), rmse(out))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [hjr.MODE_DEFAULT, hjr.MODE_DENOISE, hjr.MODE_DENOISE_UPSCALE2X])
def test_gpu_denoise_bitexact_vs_oracle(mode):
    s, oc, oa, on = _noisy_frame(w=70, h=37, spp=2)  # ragged against the 64 x 4 launch tiles
    d = s.device()
    try:
        got = d.denoise(mode, oc, oa, on)
        # and fused with the render: AOVs never leave the device
        p = s.hjr_params(70, 37, 2)
        fused = d.render_denoised(p, mode)
    finally:
        d.close()
    exp = ob.denoise(mode, oc, oa, on)
    assert got.shape == exp.shape
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "%d values differ" % int(np.sum(got.view(np.uint32) != exp.view(np.uint32)))
    assert np.array_equal(fused.view(np.uint32), exp.view(np.uint32))


@pytest.mark.gpu
def test_gpu_denoise_argument_checks():
    s = Cornell()
    d = s.device()
    try:
        img = np.zeros((8, 8, 4), np.float32)
        with pytest.raises(RuntimeError):
            d.denoise(hjr.MODE_DENOISE, img)  # guides missing
        with pytest.raises(RuntimeError):
            d.denoise(7, img, img, img)
    finally:
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode_name,mode", [("Denoise", 1), ("DenoiseUpScale2X", 2)])
def test_cli_denoise_modes(tmp_path, mode_name, mode):
    """Render_mode through the file-level drop-in: PNG == oracle render at the input size -> oracle filter -> output stage."""
    cli = os.path.join(ROOT, "henjou-renderer_amd", "henjou_cli")
    work = tmp_path / "run"
    shutil.copytree(os.path.join(hjr.ASSETS, "Model"), work / "Model")
    ro = json.load(open(os.path.join(hjr.ASSETS, "render_option_c1.json")))
    ro["Image"].update(image_width=96, image_height=64, max_spp=3, image_name="dn")
    ro["Animation"].update(start_frame=1, end_frame=2)
    ro["Render_mode"] = mode_name
    (work / "render_option.json").write_text(json.dumps(ro))
    p = subprocess.run([cli, "render_option.json"], cwd=work, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = hjr.load_png(str(work / "dn_001.png"))
    assert got.shape == (64, 96, 4)
    iw, ih = (48, 32) if mode == 2 else (96, 64)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        opt = hjr.load_render_option("render_option.json")
        sc = hjr.Scene(opt.gltf_path.decode(), opt.gltf_name.decode(), opt)
    finally:
        os.chdir(cwd)
    t = f32_time(1, opt.fps)
    cam = sc.camera(opt, t)
    op = ob.make_params(iw, ih, 3, cam.as_dict(), frame=1, seed=opt.seed, sky=tuple(opt.scene_sky_default), ibl_intensity=opt.IBL_intensity)
    oc, oa, on, _ = ob.OracleScene(sc.arrays(t), ob.MATH_PORTABLE).render(op)
    exp = hjr.float4_to_srgb8(ob.denoise(mode, oc, oa, on))[::-1]
    assert np.array_equal(got, exp), "%d pixels differ" % int(np.sum(np.any(got != exp, axis=-1)))
