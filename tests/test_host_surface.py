"""Host-side scene surface of libhenjou_hip.so (no GPU): render_option.json, glTF loader, animation/camera evaluation,
per-frame transforms — against an independent numpy restatement and hand-derived values."""
import json
import os
import shutil

import numpy as np
import pytest

import gltf_ref
from scene_util import Cornell, hjr

ASSETS = hjr.ASSETS


def test_render_option_fields():
    cwd = os.getcwd()
    os.chdir(ASSETS)
    try:
        o = hjr.load_render_option("render_option_c1.json")
    finally:
        os.chdir(cwd)
    assert (o.image_width, o.image_height, o.max_spp) == (256, 256, 16)
    assert o.image_name == b"cornelbox_c1" and o.gltf_name == b"cornelbox.gltf"
    assert o.gltf_path == b"./Model/test_gltf/"
    assert o.render_mode == 0 and o.allow_camera_animation == 1
    assert (o.fps, o.start_frame, o.end_frame) == (24, 1, 2)
    assert np.float32(o.camera_fov) == np.float32(np.pi * 45.0 / 180.0)  # degrees -> radians (render_json_loader.h:144)
    assert list(o.camera_position) == [0.0, 1.0, -7.0] and list(o.camera_direction) == [0.0, 0.0, 1.0]
    assert [np.float32(x) for x in o.scene_sky_default] == [np.float32(0.8)] * 3
    assert o.use_IBL == 0 and o.IBL_intensity == 1.0 and o.LUT_path == b"./LUT/Thin_Film_LUT.png"
    assert o.seed == 1 and o.integrator == hjr.INTEGRATOR_NEE and o.camera_animation_id == -1


def test_render_option_errors_and_fps_override(tmp_path):
    src = json.load(open(os.path.join(ASSETS, "render_option_c1.json")))
    with pytest.raises(hjr.HjrError):
        hjr.load_render_option(str(tmp_path / "missing.json"))
    # every key is mandatory (render_json_loader.h:222-225: any exception -> false)
    for sect, key in [("Image", "max_spp"), ("Camera", "camera_fov"), ("Sky", "scene_sky_default"), ("LUT", "LUT_path"),
                      ("Animation", "fps"), ("Option", "use_date")]:
        bad = json.loads(json.dumps(src))
        del bad[sect][key]
        p = tmp_path / "bad.json"
        p.write_text(json.dumps(bad))
        with pytest.raises(hjr.HjrError):
            hjr.load_render_option(str(p))
    (tmp_path / "trunc.json").write_text('{"Image": {"image_width": 12')
    with pytest.raises(hjr.HjrError):
        hjr.load_render_option(str(tmp_path / "trunc.json"))
    # unknown Render_mode -> Default; Henjou_HIP extension section; ./fps.txt overrides Animation.fps
    ok = json.loads(json.dumps(src))
    ok["Render_mode"] = "Whatever"
    ok["Henjou_HIP"] = {"seed": 77, "integrator": "MIS"}
    (tmp_path / "ok.json").write_text(json.dumps(ok))
    (tmp_path / "fps.txt").write_text("30\n")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        o = hjr.load_render_option("ok.json")
    finally:
        os.chdir(cwd)
    assert o.render_mode == 0 and o.seed == 77 and o.integrator == hjr.INTEGRATOR_MIS and o.fps == 30
    # Option.save_renderOption: a timestamped copy of the JSON text lands in the CWD (render_json_loader.h:204-219)
    ok["Option"]["save_renderOption"] = True
    (tmp_path / "save.json").write_text(json.dumps(ok))
    os.chdir(tmp_path)
    try:
        hjr.load_render_option("save.json")
    finally:
        os.chdir(cwd)
    copies = [f for f in os.listdir(tmp_path) if f.startswith("renderoption") and f.endswith(".json")]
    assert len(copies) == 1 and json.load(open(tmp_path / copies[0])) == ok


def test_gltf_loader_matches_independent_reader():
    c = Cornell()
    ref = gltf_ref.load(os.path.join(ASSETS, "Model", "test_gltf"), "cornelbox.gltf")
    a = c.arrays
    v = c.scene.view
    assert (v.n_triangles, v.n_instances, v.n_materials, v.n_lights, v.n_animations) == (984, 4, 6, 2, 5)
    assert np.array_equal(a["vertices"].reshape(-1, 3), ref["vertices"])
    assert np.array_equal(a["normals"].reshape(-1, 3), ref["normals"])
    assert np.array_equal(a["texcoords"].reshape(-1, 2), ref["texcoords"])
    assert np.array_equal(a["indices"], np.arange(984 * 3, dtype=np.uint32))  # de-indexed (gltfloader.h:1491)
    assert np.array_equal(a["material_ids"], ref["material_ids"])
    assert np.array_equal(a["prim_offsets"], ref["prim_offsets"]) and list(a["prim_offsets"]) == [0, 10, 12, 972]
    assert np.array_equal(a["light_prim_ids"], ref["light_prim_ids"]) and list(a["light_prim_ids"]) == [10, 11]
    assert np.array_equal(a["light_prim_emission"].reshape(-1, 3), ref["light_prim_emission"])
    assert np.array_equal(a["instance_animation_id"], ref["instance_animation_id"])
    assert np.array_equal(a["geometry_index_offset"], a["prim_offsets"] * 3)
    for i, m in enumerate(ref["materials"]):
        got = a["materials"][i]
        for k in ("basecolor", "emission"):
            assert np.array_equal(got[k], m[k]), (i, k)
        for k in ("metallic", "roughness", "sheen", "clearcoat", "ior", "transmission", "is_light", "ideal_specular", "is_thinfilm"):
            assert got[k] == m[k], (i, k, got[k], m[k])
    # tinygltf defaults the reference relies on: missing metallicFactor -> 1.0, emissive strength folded into emission
    assert a["materials"][5]["metallic"] == 1.0 and a["materials"][4]["ideal_specular"] == 1
    assert list(a["materials"][3]["emission"]) == [10.0, 10.0, 10.0] and a["materials"][3]["is_light"] == 1
    # camera node override (gltfloader.h:1514-1522)
    assert c.opt.camera_animation_id == ref["camera"]["animation_id"] == 0
    assert np.float32(c.opt.camera_fov) == ref["camera"]["fov"]
    assert list(c.opt.camera_position) == [0, 0, 0] and list(c.opt.camera_direction) == [0, 0, -1]


def test_camera_and_transforms():
    c = Cornell()
    cam = c.camera.as_dict()
    # camera_f = 2 / tan(fov) with the FULL angle (renderer.h:1147)
    assert np.float32(cam["f"]) == np.float32(2.0 / np.tan(np.float32(c.opt.camera_fov)))
    g = json.load(open(os.path.join(ASSETS, "Model", "test_gltf", "cornelbox.gltf")))
    assert np.allclose(cam["pos"], g["nodes"][0]["translation"], atol=0, rtol=1e-7)
    assert np.allclose(cam["dir"], [-1, 0, 0], atol=1e-6) and np.allclose(cam["up"], [0, 1, 0], atol=1e-6)
    assert np.allclose(cam["right"], [0, 0, -1], atol=1e-6)
    m, inv = c.arrays["transforms"], c.arrays["inv_transforms"]
    assert np.array_equal(m[0], np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32))
    # node 2: rotation (0,1,0,0) = 180 deg about Y, negative scale, translation y (T * R * S, no hierarchy: animation.h:81-94)
    s = np.array(g["nodes"][2]["scale"], np.float32)
    assert np.allclose(m[1].reshape(3, 4)[:, :3], np.diag([-s[0], s[1], -s[2]]), atol=1e-7)
    assert np.allclose(m[1].reshape(3, 4)[:, 3], g["nodes"][2]["translation"], atol=1e-7)
    for i in range(4):
        M = np.vstack([m[i].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        Mi = np.vstack([inv[i].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        assert np.allclose(M @ Mi, np.eye(4), atol=1e-6)
    # time before the first key / static camera branch (renderer.h:1163-1168)
    c.opt.allow_camera_animation = 0
    c.opt.camera_position = (hjr.C.c_float * 3)(1, 2, 3)
    c.opt.camera_direction = (hjr.C.c_float * 3)(0, 0, 2)
    st = c.scene.camera(c.opt, 0.0).as_dict()
    assert st["pos"] == [1, 2, 3] and st["right"] == [-2.0, 0.0, 0.0] and st["up"] == [0.0, 4.0, 0.0]  # un-normalised, sic


def test_animation_interpolation_between_keys(tmp_path):
    """LINEAR lerp between appended keys, sampler chosen by channel index, quaternion not re-normalised."""
    d = tmp_path / "m"
    shutil.copytree(os.path.join(ASSETS, "Model", "test_gltf"), d)
    g = json.load(open(d / "cornelbox.gltf"))
    raw = bytearray(open(d / "cornelbox.bin", "rb").read())
    # animate node 3 (sphere) translation: keys (1/24, 5) -> values (0,0,0) and (10,0,0)
    off = len(raw)
    raw += np.array([0, 0, 0, 10, 0, 0], np.float32).tobytes()
    g["buffers"][0]["byteLength"] = len(raw)
    g["bufferViews"].append({"buffer": 0, "byteLength": 24, "byteOffset": off})
    g["accessors"].append({"bufferView": len(g["bufferViews"]) - 1, "componentType": 5126, "count": 2, "type": "VEC3"})
    g["animations"].append({"channels": [{"sampler": 0, "target": {"node": 3, "path": "translation"}}],
                            "samplers": [{"input": 24, "interpolation": "LINEAR", "output": len(g["accessors"]) - 1}]})
    open(d / "cornelbox.gltf", "w").write(json.dumps(g))
    open(d / "cornelbox.bin", "wb").write(bytes(raw))
    opt = hjr.load_render_option(os.path.join(ASSETS, "render_option_c1.json"))
    sc = hjr.Scene(str(d), "cornelbox.gltf", opt)
    k0, k1 = np.float32(1 / 24), np.float32(5)
    t = np.float32(2.5)
    delta = np.float32(t - k0) / np.float32(k1 - k0)
    m, _ = sc.transforms(float(t))
    expect_x = np.float32(0) * (np.float32(1) - delta) + np.float32(10) * delta
    assert np.float32(m[2][3]) == expect_x
    m0, _ = sc.transforms(0.0)  # time 0: key 0 = the node's own TRS (gltfloader.h:1313-1343)
    assert np.float32(m0[2][3]) == np.float32(g["nodes"][3]["translation"][0]) * 0 + np.float32(m0[2][3])
    m_end, _ = sc.transforms(100.0)  # past the last key: last value (animation.h:58)
    assert m_end[2][3] == 10.0


def test_gltf_error_paths(tmp_path):
    opt = hjr.load_render_option(os.path.join(ASSETS, "render_option_c1.json"))
    with pytest.raises(hjr.HjrError):
        hjr.Scene(str(tmp_path), "nope.gltf", opt)
    (tmp_path / "bad.gltf").write_text("{ not json")
    with pytest.raises(hjr.HjrError):
        hjr.Scene(str(tmp_path), "bad.gltf", opt)
    # index pointing past the vertex accessor -> the arrayAdapter's out_of_range (gltfloader.h:942-950)
    d = tmp_path / "m"
    shutil.copytree(os.path.join(ASSETS, "Model", "test_gltf"), d)
    g = json.load(open(d / "cornelbox.gltf"))
    g["accessors"][0]["count"] = 3
    open(d / "cornelbox.gltf", "w").write(json.dumps(g))
    with pytest.raises(hjr.HjrError):
        hjr.Scene(str(d), "cornelbox.gltf", opt)


def test_glb_container(tmp_path):
    """Binary glTF (gltfloader.h:1086-1091): JSON chunk + BIN chunk; must load to the same SceneData as the .gltf."""
    import struct
    src = os.path.join(ASSETS, "Model", "test_gltf")
    g = json.load(open(os.path.join(src, "cornelbox.gltf")))
    binary = open(os.path.join(src, "cornelbox.bin"), "rb").read()
    del g["buffers"][0]["uri"]
    js = json.dumps(g).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    binary += b"\0" * ((4 - len(binary) % 4) % 4)
    total = 12 + 8 + len(js) + 8 + len(binary)
    blob = struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(binary), 0x004E4942) + binary
    (tmp_path / "c.glb").write_bytes(blob)
    opt = hjr.load_render_option(os.path.join(ASSETS, "render_option_c1.json"))
    a = hjr.Scene(str(tmp_path), "c.glb", opt).arrays()
    b = Cornell().arrays
    for k in ("vertices", "normals", "texcoords", "material_ids", "prim_offsets", "light_prim_ids"):
        assert np.array_equal(a[k], b[k]), k
    (tmp_path / "bad.glb").write_bytes(blob[:40])
    with pytest.raises(hjr.HjrError):
        hjr.Scene(str(tmp_path), "bad.glb", opt)


def test_reference_shipped_config_verbatim(tmp_path):
    """The one piece of reference-held DATA on this boundary: the render_option.json + fps.txt the reference ships in its working
    directory (committed verbatim under tests/golden/ref_config/), parsed by the product's loader; every field against the value
    written in the file (render_json_loader.h:97-202), then the file-level entry point on it: the glTF it names is not in the
    reference repository, so hjr_render_file must fail with HJR_ERR_IO naming that file — before any GPU call."""
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_config")
    for f in ("render_option.json", "fps.txt"):
        shutil.copy(os.path.join(ref, f), str(tmp_path / f))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))  # load_json reads ./fps.txt of the working directory (render_json_loader.h:164)
    try:
        o = hjr.load_render_option("render_option.json")
        rc = hjr.lib().hjr_render_file(b"render_option.json", 0)
        err = hjr.lib().hjr_last_error().decode()
    finally:
        os.chdir(cwd)
    assert (o.image_width, o.image_height, o.max_spp) == (1280, 720, 5000)
    assert o.image_name == b"multiple_scattering_ggx" and o.image_directory == b"./"
    assert o.render_mode == hjr.MODE_DEFAULT
    assert o.gltf_path == b"./Model/White_FurnanceTest/GLTF/" and o.gltf_name == b"WhiteFurnanceTest_Roghness.gltf"
    assert o.allow_camera_animation == 1
    assert list(o.camera_position) == [0.0, 1.0, -7.0] and list(o.camera_direction) == [0.0, 0.0, 1.0]
    assert np.float32(o.camera_fov) == np.float32(np.pi * 45.0 / 180.0)  # 45 degrees -> radians (render_json_loader.h:144)
    assert o.ptxfile_path == b"../lib/ptx/Release/HenjouRenderer_generated_henjouRendererCU.cu.optixir"
    assert (o.fps, o.start_frame, o.end_frame) == (24, 1, 2) and o.time_limit == 5.0  # fps.txt holds 24 as well
    assert o.IBL_path == b"./HDRI/blocky_photo_studio_4k.hdr" and o.IBL_intensity == 1.0 and o.use_IBL == 0
    assert [np.float32(x) for x in o.scene_sky_default] == [np.float32(0.8)] * 3
    assert o.use_date == 1 and o.save_renderOption == 0
    assert o.LUT_path == b"./LUT/Thin_Film_LUT.png"
    assert (o.seed, o.integrator, o.devices, o.tile) == (1, hjr.INTEGRATOR_NEE, 1, 8)  # no Henjou_HIP section: defaults
    assert rc == -2, (rc, err)  # HJR_ERR_IO
    assert "WhiteFurnanceTest_Roghness.gltf" in err
    # the override file is honoured: another rate in ./fps.txt wins over the JSON's (render_json_loader.h:14-34, 164-171)
    (tmp_path / "fps.txt").write_text("30")
    os.chdir(str(tmp_path))
    try:
        assert hjr.load_render_option("render_option.json").fps == 30
    finally:
        os.chdir(cwd)
