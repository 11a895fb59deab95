"""§8 row f2 (CPU part): texture slots through the glTF loader, 8-bit texture sampling rules, Radiance HDR reader and the
equirect sky lookup of the oracle; the GPU parity for the same features is in tests/test_gpu_scenes.py."""
import os
import struct

import numpy as np
import pytest

import oracle_binding as ob
from scene_util import Cornell, hjr

L = ob.lib()


def tex_fetch(img, srgb, u, v):
    o = ob.F3()
    L.hjo_tex_fetch(img.ctypes.data, img.shape[1], img.shape[0], srgb, u, v, o)
    return np.array(o, np.float32)


def test_loader_binds_texture_slots():
    c = Cornell("render_option_tex.json")
    v = c.scene.view
    assert v.n_textures == 1 and v.n_materials == 5
    mats = c.arrays["materials"]
    assert mats[4]["basecolor_tex"] == 0 and all(mats[i]["basecolor_tex"] == -1 for i in range(4))
    assert all(m["metallic_roughness_tex"] == -1 and m["normal_tex"] == -1 and m["emission_tex"] == -1 for m in mats)
    assert list(mats[4]["basecolor"]) == [1.0, 1.0, 1.0]  # tinygltf default factor
    px, srgb = c.arrays["textures"][0]
    from PIL import Image
    ref = np.array(Image.open(os.path.join(hjr.ASSETS, "Model", "test_gltf", "texture", "Tex.png")).convert("RGBA"))
    assert srgb == 1 and np.array_equal(px, ref)
    # the non-animated camera branch (renderer.h:1163-1168)
    assert c.opt.camera_animation_id == -1 and np.allclose(c.camera.as_dict()["dir"], [-1, 0, 0])


def test_texture_sampling_rules():
    img = np.zeros((2, 4, 4), np.uint8)
    img[..., 0] = [[0, 64, 128, 255], [255, 128, 64, 0]]
    img[..., 1] = 200
    img[..., 3] = 255
    # texel centres, linear (NonColor) and sRGB-decoded reads
    assert np.isclose(tex_fetch(img, 0, 1.5 / 4, 0.5 / 2)[0], 64 / 255, atol=1e-7)
    s = 128 / 255
    assert np.isclose(tex_fetch(img, 1, 2.5 / 4, 0.25)[0], ((s + 0.055) / 1.055) ** 2.4, atol=1e-6)
    assert np.isclose(tex_fetch(img, 1, 0.5 / 4, 0.25)[0], 0.0) and np.isclose(tex_fetch(img, 1, 3.5 / 4, 0.25)[0], 1.0, atol=1e-7)
    # decode happens BEFORE filtering: halfway between texels 64 and 128 is the mean of the decoded values
    a, b = tex_fetch(img, 1, 1.5 / 4, 0.25)[0], tex_fetch(img, 1, 2.5 / 4, 0.25)[0]
    assert np.isclose(tex_fetch(img, 1, 2.0 / 4, 0.25)[0], 0.5 * (a + b), atol=1e-6)
    # wrap in both directions, row 0 is the top row (v = 0)
    assert np.allclose(tex_fetch(img, 0, 0.5 / 4 + 3.0, 0.25 - 1.0), tex_fetch(img, 0, 0.5 / 4, 0.25))
    assert np.isclose(tex_fetch(img, 0, 0.5 / 4, 0.75)[0], 1.0, atol=1e-7)


def write_hdr(path, img, rle):
    """img: float [h, w, 3] -> Radiance RGBE file (flat or new-style RLE with literal runs only)."""
    h, w, _ = img.shape
    m = img.max(axis=-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0).astype(int)
    scale = np.where(m > 1e-32, 256.0 / np.exp2(e), 0.0)
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, e + 128, 0)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            if not rle:
                f.write(rgbe[y].tobytes())
            else:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for c in range(4):
                    x = 0
                    while x < w:
                        n = min(128, w - x)
                        f.write(bytes([n]) + rgbe[y, x:x + n, c].tobytes())
                        x += n
    sc = np.where(rgbe[..., 3:4] > 0, np.exp2(rgbe[..., 3:4].astype(np.float64) - 136.0), 0.0)
    return (rgbe[..., :3] * sc).astype(np.float32)


@pytest.mark.parametrize("rle,w", [(False, 16), (True, 16), (True, 300)])
def test_radiance_hdr_reader(tmp_path, rle, w):
    rng = np.random.default_rng(5)
    img = (rng.uniform(0, 1, (6, w, 3)) ** 4 * 50).astype(np.float32)
    img[0, 0] = 0
    exp = write_hdr(str(tmp_path / "t.hdr"), img, rle)
    got = hjr.load_hdr(str(tmp_path / "t.hdr"))
    assert got.shape == (6, w, 4) and np.array_equal(got[..., :3], exp) and (got[..., 3] == 0).all()
    with pytest.raises(hjr.HjrError):
        hjr.load_hdr(str(tmp_path / "missing.hdr"))
    (tmp_path / "junk.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\nxx")
    with pytest.raises(hjr.HjrError):
        hjr.load_hdr(str(tmp_path / "junk.hdr"))


def test_equirect_sky_lookup():
    h, w = 8, 16
    sky = np.zeros((h, w, 4), np.float32)
    sky[..., 0] = np.arange(w)[None, :]
    sky[..., 1] = np.arange(h)[:, None]
    for mode in (ob.MATH_PORTABLE, ob.MATH_LIBM):
        def f(d):
            o = ob.F3()
            L.hjo_sky_fetch(mode, sky.ctypes.data, w, h, ob.F3(*d), o)
            return np.array(o)
        up = f((0, 1, 0))          # v = acos(1)/pi = 0 -> blends row 7 and row 0 (wrap)
        assert np.isclose(up[1], 3.5, atol=1e-5)
        horizon = f((1, 0, 0))     # u = 0.5, v = 0.5 -> texel boundary (8, 4): mean of columns 7/8 and rows 3/4
        assert np.isclose(horizon[0], 7.5, atol=1e-4) and np.isclose(horizon[1], 3.5, atol=1e-4)
        assert np.isclose(f((0, 0, 1))[0], 0.75 * w - 0.5, atol=1e-4)   # +z: u = 0.75
        assert np.isclose(f((0, 0, -1))[0], 0.25 * w - 0.5, atol=1e-4)  # -z: u = 0.25
    xs = np.random.default_rng(9).normal(size=(2000, 2)).astype(np.float32)
    got = np.array([L.hjo_p_atan2(float(a), float(b)) for a, b in xs])
    assert np.max(np.abs(got - np.arctan2(xs[:, 0].astype(np.float64), xs[:, 1]))) < 5e-7
    assert L.hjo_p_atan2(0.0, 1.0) == 0.0 and abs(L.hjo_p_atan2(0.0, -1.0) - np.pi) < 1e-6 and abs(L.hjo_p_atan2(-1.0, 0.0) + np.pi / 2) < 1e-6


def test_oracle_renders_textured_scene_and_sky():
    c = Cornell("render_option_tex.json")
    osc = ob.OracleScene(c.arrays, ob.MATH_PORTABLE)
    img, alb, _, st = osc.render(c.oracle_params(64, 64, 4))
    assert st["nan_samples"] == 0 and np.isfinite(img).all()
    # the albedo AOV of the textured box varies across pixels (a constant factor would not)
    assert len(np.unique(np.round(alb[..., 0], 4))) > 20
    sky = np.zeros((4, 8, 4), np.float32)
    sky[..., 2] = 5.0
    arrays = dict(Cornell().arrays, sky_rgba=sky)
    a, _, _, _ = ob.OracleScene(arrays, ob.MATH_PORTABLE).render(c.oracle_params(48, 48, 2))
    b, _, _, _ = ob.OracleScene(Cornell().arrays, ob.MATH_PORTABLE).render(c.oracle_params(48, 48, 2))
    assert not np.array_equal(a, b)


def test_oracle_normal_map(tmp_path):
    """Material.normal_tex through the loader (NonColor slot) and the oracle's build-defined tangent-space lookup: the normal AOV of the
    mapped surfaces changes, the albedo AOV does not, normals stay unit length where a map applies."""
    from scene_util import make_normal_mapped_scene
    c, n_mapped = make_normal_mapped_scene(tmp_path)
    mats = c.arrays["materials"]
    assert c.scene.view.n_textures == 2 and sum(int(m["normal_tex"] >= 0) for m in mats) == n_mapped
    slot = int(mats[0]["normal_tex"])
    assert c.arrays["textures"][slot][1] == 0  # the normal map is a NonColor texture (gltfloader.h:1171)
    plain = Cornell("render_option_tex.json")
    a_img, a_alb, a_nrm, st = ob.OracleScene(c.arrays, ob.MATH_PORTABLE).render(c.oracle_params(64, 48, 1))
    b_img, b_alb, b_nrm, _ = ob.OracleScene(plain.arrays, ob.MATH_PORTABLE).render(plain.oracle_params(64, 48, 1))
    assert st["nan_samples"] == 0 and np.isfinite(a_img).all()
    assert np.array_equal(a_alb, b_alb) and not np.array_equal(a_nrm, b_nrm)
    changed = np.any(a_nrm[..., :3] != b_nrm[..., :3], axis=-1)
    assert 0.05 < changed.mean() < 0.95
    assert np.allclose(np.linalg.norm(a_nrm[changed][:, :3], axis=-1), 1.0, atol=1e-5)
