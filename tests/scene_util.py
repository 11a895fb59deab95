"""Shared helpers for the tests: loading the bundled scene through the product's loader, oracle params."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

hjr = entry.load_package()


def f32_time(frame, fps):
    return float(np.float32(frame) / np.float32(fps))


DEVICE_OPTIONS = {}  # options every Cornell.device() applies (tests force kernel layouts / families with them)


class device_options:
    """with device_options(lds_bvh=0, pipeline="wf"): ...  — hjr_set_option values for the devices created inside (the library reads no
    environment variable).  pipeline takes "mega" / "wf" or the numbers 1 / 2."""

    def __init__(self, **kv):
        self.kv = {k: ({"mega": 1, "wf": 2}.get(v, v) if k == "pipeline" else v) for k, v in kv.items()}

    def __enter__(self):
        self.old = dict(DEVICE_OPTIONS)
        DEVICE_OPTIONS.update(self.kv)
        return self

    def __exit__(self, *a):
        DEVICE_OPTIONS.clear()
        DEVICE_OPTIONS.update(self.old)
        return False


def new_device(options=None):
    """hjr.Device(0) with the DEVICE_OPTIONS of the enclosing `device_options` blocks (and `options`) applied."""
    d = hjr.Device(0)
    for k, v in dict(DEVICE_OPTIONS, **(options or {})).items():
        d.set_option(k, v)
    return d


class Cornell:
    """cornelbox.gltf loaded via libhenjou_hip.so's scene surface, at frame 1 (t = 1/24 s)."""

    def __init__(self, config="render_option_c1.json", gltf=None):
        cwd = os.getcwd()
        os.chdir(hjr.ASSETS)
        try:
            self.opt = hjr.load_render_option(config)
            if gltf:
                self.opt.gltf_name = gltf.encode()
            self.scene = hjr.Scene(self.opt.gltf_path.decode(), self.opt.gltf_name.decode(), self.opt)
        finally:
            os.chdir(cwd)
        self.time = f32_time(1, self.opt.fps)
        self.camera = self.scene.camera(self.opt, self.time)
        self.arrays = self.scene.arrays(self.time)

    def hjr_params(self, w, h, spp, **kw):
        kw.setdefault("sky", tuple(self.opt.scene_sky_default))
        kw.setdefault("ibl_intensity", self.opt.IBL_intensity)
        return hjr.make_params(w, h, spp, self.camera, **kw)

    def oracle_params(self, w, h, spp, **kw):
        import oracle_binding as ob
        kw.setdefault("sky", tuple(self.opt.scene_sky_default))
        kw.setdefault("ibl_intensity", self.opt.IBL_intensity)
        return ob.make_params(w, h, spp, self.camera.as_dict(), **kw)

    def device(self, options=None):
        """A device context with this scene resident.  `options` (and the module-wide DEVICE_OPTIONS the `device_options` context manager
        sets) go through hjr_set_option BEFORE the frame data is built, so that layout options act on it."""
        d = new_device(options)
        d.upload_scene(self.scene.view)
        d.set_transforms(self.arrays["transforms"], self.arrays["inv_transforms"])
        return d


def load_lut():
    return hjr.load_png(os.path.join(hjr.ASSETS, "LUT", "Thin_Film_LUT.png"))


class StressScene(Cornell):
    """tools/make_stress_scene.py output loaded through the product's scene surface."""

    def __init__(self, outdir, spheres=8, segments=32):
        import subprocess
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_stress_scene.py"), str(outdir),
                               "--spheres", str(spheres), "--segments", str(segments)], stdout=subprocess.DEVNULL)
        self.opt = hjr.load_render_option(os.path.join(str(outdir), "render_option_stress.json"))
        self.scene = hjr.Scene(self.opt.gltf_path.decode(), self.opt.gltf_name.decode(), self.opt)
        self.time = f32_time(1, self.opt.fps)
        self.camera = self.scene.camera(self.opt, self.time)
        self.arrays = self.scene.arrays(self.time)


def make_normal_mapped_scene(tmp_path):
    """cornelbox_texture_test.gltf with a generated wavy tangent-space normal map bound to the textured box and to material 0;
    returns (Cornell, number of materials with a normal map)."""
    import json
    import shutil
    from PIL import Image
    work = tmp_path / "nm"
    shutil.copytree(os.path.join(hjr.ASSETS, "Model"), work / "Model")
    yy, xx = np.mgrid[0:64, 0:64]
    nx = 0.35 * np.sin(xx * 0.5) * np.cos(yy * 0.3)
    ny = 0.35 * np.cos(xx * 0.2 + yy * 0.4)
    nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
    img = np.stack([(nx * 0.5 + 0.5) * 255, (ny * 0.5 + 0.5) * 255, (nz * 0.5 + 0.5) * 255], -1).round().astype(np.uint8)
    Image.fromarray(img).save(str(work / "Model" / "test_gltf" / "texture" / "Normal.png"))
    gpath = work / "Model" / "test_gltf" / "cornelbox_texture_test.gltf"
    g = json.load(open(gpath))
    g["images"].append({"uri": "texture/Normal.png"})
    g["textures"].append({"source": len(g["images"]) - 1})
    tex_index = len(g["textures"]) - 1
    mapped = sorted(set([i for i, m in enumerate(g["materials"]) if "baseColorTexture" in m.get("pbrMetallicRoughness", {})] + [0]))
    for i in mapped:
        g["materials"][i]["normalTexture"] = {"index": tex_index}
    json.dump(g, open(gpath, "w"))
    ro = json.load(open(os.path.join(hjr.ASSETS, "render_option_tex.json")))
    ro["GLTF_file"]["gltf_filepath"] = str(work / "Model" / "test_gltf") + "/"
    (work / "render_option.json").write_text(json.dumps(ro))
    return Cornell(str(work / "render_option.json")), len(mapped)
