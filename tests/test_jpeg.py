"""Baseline JPEG texture decoder (host/jpeg.cpp).  The reference decodes with stb_image, which is not available here; an
independent decoder (Pillow / libjpeg) differs from any other conforming decoder by a few LSB (IDCT precision, chroma
upsampling rounding, colour-conversion fixed point), so the check is: within 3 LSB everywhere, mean difference < 0.5 LSB,
and exact where no lossy arithmetic is involved (flat blocks of a grey image at quality 100)."""
import io
import json
import os
import shutil

import numpy as np
import pytest
from PIL import Image

from scene_util import hjr

rng = np.random.default_rng(11)


def _photo(h, w):
    """Smooth + detailed RGB test image."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(x / 9.0) * np.cos(y / 13.0), 127 + 100 * np.cos((x + y) / 17.0), 40 + 0.9 * (x * 255 / max(w - 1, 1))], axis=-1)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(64, 64), (37, 53), (8, 8), (1, 1), (130, 17)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])  # 4:4:4, 4:2:2, 4:2:0
def test_rgb_against_pillow(tmp_path, size, subsampling):
    h, w = size
    img = _photo(h, w)
    p = str(tmp_path / "t.jpg")
    Image.fromarray(img).save(p, quality=90, subsampling=subsampling)
    got = hjr.load_image(p)
    exp = np.array(Image.open(p).convert("RGB"))
    assert got.shape == (h, w, 4) and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3].astype(int) - exp.astype(int))
    if subsampling == 1 and w > 2:
        # stb_image's 2x horizontal upsampler weights the last chroma pair as (3 * in[w-2] + in[w-1]) where libjpeg uses
        # (in[w-2] + 3 * in[w-1]): the decoder keeps stb's published arithmetic, so the last two columns are not compared
        d = d[:, :-2]
    assert d.max() <= 3, d.max()
    assert d.mean() < 0.5, d.mean()


def test_grey_restart_and_flat_blocks(tmp_path):
    g = np.zeros((40, 56), np.uint8)
    g[:24] = 200
    g[24:, :16] = 17
    p = str(tmp_path / "g.jpg")
    Image.fromarray(g, "L").save(p, quality=100)
    got = hjr.load_image(p)
    assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()
    assert np.array_equal(got[..., 0], np.array(Image.open(p)))  # DC-only blocks: no rounding freedom
    # restart intervals
    img = _photo(48, 80)
    q = str(tmp_path / "r.jpg")
    try:
        Image.fromarray(img).save(q, quality=85, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("Pillow without restart_marker_blocks")
    raw = open(q, "rb").read()
    if b"\xff\xdd" not in raw:
        pytest.skip("encoder wrote no DRI segment")
    got = hjr.load_image(q)
    exp = np.array(Image.open(q).convert("RGB"))
    assert np.abs(got[..., :3].astype(int) - exp.astype(int)).max() <= 3


def test_rejects_progressive_and_garbage(tmp_path):
    p = str(tmp_path / "p.jpg")
    Image.fromarray(_photo(32, 32)).save(p, quality=80, progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        hjr.load_image(p)
    q = str(tmp_path / "x.jpg")
    open(q, "wb").write(b"\xff\xd8\xff\xdb\x00\x05garbage")
    with pytest.raises(RuntimeError):
        hjr.load_image(q)
    # PNG still goes through the same front end
    r = str(tmp_path / "a.png")
    Image.fromarray(_photo(9, 7)).save(r)
    assert np.array_equal(hjr.load_image(r)[..., :3], np.array(Image.open(r)))


def test_gltf_with_jpeg_texture(tmp_path):
    """cornelbox_texture_test.gltf with its PNG re-encoded as JPEG: the loader binds it like the PNG."""
    src = os.path.join(hjr.ASSETS, "Model", "test_gltf")
    dst = tmp_path / "Model" / "test_gltf"
    shutil.copytree(src, dst)
    g = json.load(open(dst / "cornelbox_texture_test.gltf"))
    assert len(g["images"]) == 1
    png = dst / g["images"][0]["uri"]
    jpg = str(png)[:-4] + ".jpg"
    Image.open(png).convert("RGB").save(jpg, quality=92)
    g["images"][0]["uri"] = g["images"][0]["uri"][:-4] + ".jpg"
    json.dump(g, open(dst / "cornelbox_texture_test.gltf", "w"))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        opt = hjr.load_render_option(os.path.join(hjr.ASSETS, "render_option_tex.json"))
        sc = hjr.Scene("./Model/test_gltf/", "cornelbox_texture_test.gltf", opt)
    finally:
        os.chdir(cwd)
    assert sc.view.n_textures == 1
    import ctypes as C
    t = C.cast(sc.view.textures, C.POINTER(hjr.Texture))[0]
    ref = np.array(Image.open(jpg).convert("RGB"))
    assert (t.width, t.height) == (ref.shape[1], ref.shape[0])
    px = np.ctypeslib.as_array(C.cast(t.rgba8, C.POINTER(C.c_uint8)), shape=(t.height, t.width, 4))
    assert np.abs(px[..., :3].astype(int) - ref.astype(int)).max() <= 3


def test_corrupt_streams_do_not_crash(tmp_path):
    """Random byte flips / truncations of valid files: an error or an image, never a fault (run in a child process)."""
    import subprocess
    import sys
    src = tmp_path / "seed"
    src.mkdir()
    Image.fromarray(_photo(40, 56)).save(str(src / "a.jpg"), quality=85, subsampling=2)
    Image.fromarray(_photo(24, 24)).save(str(src / "b.jpg"), quality=95, subsampling=0)
    Image.fromarray(_photo(33, 20)[..., 0], "L").save(str(src / "c.jpg"), quality=70)
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from scene_util import hjr
rng = np.random.default_rng(3)
n_ok = n_err = 0
for name in ("a.jpg", "b.jpg", "c.jpg"):
    raw = bytearray(open(os.path.join(sys.argv[2], name), "rb").read())
    for trial in range(150):
        b = bytearray(raw)
        if trial % 3 == 0:
            b = b[: int(rng.integers(2, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        p = os.path.join(sys.argv[2], "m.jpg")
        open(p, "wb").write(bytes(b))
        try:
            a = hjr.load_image(p)
            assert a.ndim == 3 and a.shape[2] == 4
            n_ok += 1
        except RuntimeError:
            n_err += 1
print("ok", n_ok, "err", n_err)
'''
    from scene_util import ROOT
    out = subprocess.run([sys.executable, "-c", code, ROOT, str(src)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout
