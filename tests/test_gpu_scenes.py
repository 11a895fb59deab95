"""GPU parity on the other BASELINE.json configurations: thin-film LUT BSDF (configs[2]), negative-IOR glass with
ior 1.5 (configs[3]), the diffuse/specular-only variant (configs[1] strict reading) and the synthetic many-triangle
stress scene.  Same bar as test_gpu_parity.py: bit-exact against the oracle's PORTABLE mode."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from scene_util import new_device, Cornell, StressScene, hjr, load_lut
from test_gpu_parity import assert_bitexact

pytestmark = pytest.mark.gpu


def check(scene, w, h, spp, lut=None, integrator=hjr.INTEGRATOR_NEE):
    d = scene.device()
    try:
        arrays = dict(scene.arrays)
        if lut is not None:
            d.set_lut(lut)
            arrays["lut_rgba"] = lut
        color, albedo, normal = d.render(scene.hjr_params(w, h, spp, integrator=integrator))
        st_kernel = d.stats()
    finally:
        d.close()
    osc = ob.OracleScene(arrays, ob.MATH_PORTABLE)
    oc, oa, on, st = osc.render(scene.oracle_params(w, h, spp, integrator=integrator))
    assert st["nan_samples"] == 0
    assert_bitexact(color, oc, "aov_color")
    assert_bitexact(albedo, oa, "aov_albedo")
    assert_bitexact(normal, on, "aov_normal")
    return color, st_kernel


def test_thinfilm_lut_scene():
    lut = load_lut()
    s = Cornell("render_option_c3.json")
    assert s.arrays["materials"][0]["is_thinfilm"] == 1 and s.arrays["materials"][5]["is_thinfilm"] == 1
    with_lut, _ = check(s, 96, 64, 8, lut=lut)
    # the LUT must actually matter: same scene without the ThinFilm extension differs
    plain, _ = check(Cornell("render_option_c1.json"), 96, 64, 8)
    assert not np.array_equal(with_lut, plain)
    # a thin-film material without a bound LUT reads F0 = 0 on both sides (defined behaviour, not a crash)
    check(s, 32, 32, 2, lut=None)


def test_negative_ior_glass_ior15():
    s = Cornell("render_option_c4.json")
    assert s.arrays["materials"][4]["ior"] == np.float32(1.5) and s.arrays["materials"][4]["ideal_specular"] == 1
    a, _ = check(s, 96, 64, 8)
    b, _ = check(Cornell("render_option_c1.json"), 96, 64, 8)
    assert not np.array_equal(a, b)
    check(s, 64, 48, 4, integrator=hjr.INTEGRATOR_MIS)


def test_diffuse_specular_only_variant():
    s = Cornell("render_option_c2_nodiel.json")
    assert s.arrays["materials"][4]["ideal_specular"] == 0 and s.arrays["materials"][4]["metallic"] == 1.0
    check(s, 96, 64, 8)  # the sphere becomes roughness-0 multiple-scattering GGX (alpha clamps to 1e-4, BSDFs.h:829)


def test_stress_scene_small(tmp_path):
    s = StressScene(tmp_path, spheres=8, segments=32)
    assert s.scene.view.n_triangles == 12 + 8 * 960 and s.scene.view.n_instances == 10
    _, st = check(s, 160, 90, 4)
    assert st["bvh_depth"] < 32 and st["n_triangles"] == s.scene.view.n_triangles
    check(s, 64, 36, 2, integrator=hjr.INTEGRATOR_PT)


def test_stress_scene_100k(tmp_path):
    """~250 k triangles: deeper BVH, instance transforms with arbitrary rotations; window checked against the oracle."""
    s = StressScene(tmp_path, spheres=16, segments=128)
    assert s.scene.view.n_triangles > 250000
    check(s, 48, 27, 2)


def test_textured_scene():
    """cornelbox_texture_test.gltf (base-colour PNG texture, sRGB) — §8 row f2."""
    s = Cornell("render_option_tex.json")
    assert s.scene.view.n_textures == 1
    color, _ = check(s, 128, 96, 8)
    assert np.isfinite(color).all()


def test_equirect_sky(tmp_path):
    """IBL through the miss program: generated equirect HDR, uploaded with hjr_set_sky, against the oracle."""
    rng = np.random.default_rng(3)
    sky = np.zeros((32, 64, 4), np.float32)
    sky[..., :3] = rng.uniform(0.0, 3.0, (32, 64, 3)).astype(np.float32)
    s = Cornell()
    d = s.device()
    try:
        d.set_sky(sky)
        p = s.hjr_params(96, 64, 6, ibl_intensity=0.7)
        color, albedo, normal = d.render(p)
        d.set_sky(None)
        plain, _, _ = d.render(p)
    finally:
        d.close()
    arrays = dict(s.arrays, sky_rgba=sky)
    oc, oa, on, st = ob.OracleScene(arrays, ob.MATH_PORTABLE).render(s.oracle_params(96, 64, 6, ibl_intensity=0.7))
    assert_bitexact(color, oc, "aov_color")
    assert_bitexact(normal, on, "aov_normal")
    assert not np.array_equal(color, plain)


def test_random_materials_and_cameras():
    """Randomised sweep over the material parameter space (all Disney terms incl. sheen / clearcoat, the msGGX threshold, glass
    with various ior, thin-film on/off) and camera poses: every frame must match the oracle bit for bit."""
    import copy
    rng = np.random.default_rng(20260)
    lut = load_lut()
    base = Cornell()
    for trial in range(6):
        s = copy.copy(base)
        a = dict(base.arrays)
        mats = a["materials"].copy()
        for i in range(len(mats)):
            if mats[i]["is_light"]:
                continue
            mats[i]["basecolor"] = rng.uniform(0.05, 1.0, 3).astype(np.float32)
            mats[i]["metallic"] = np.float32(rng.choice([0.0, 0.3, 0.5, 0.51, 1.0]))
            mats[i]["roughness"] = np.float32(rng.choice([0.0, 0.05, 0.3, 0.7, 1.0]))
            mats[i]["sheen"] = np.float32(rng.choice([0.0, 0.5]))
            mats[i]["clearcoat"] = np.float32(rng.choice([0.0, 1.0]))
            mats[i]["transmission"] = np.float32(rng.choice([0.0, 1.0]))
            mats[i]["ior"] = np.float32(rng.choice([1.0, 1.33, 1.5, 2.4]))
            mats[i]["is_thinfilm"] = int(rng.integers(0, 2))
            mats[i]["ideal_specular"] = int(mats[i]["roughness"] == 0 and mats[i]["transmission"] > 0)  # gltfloader.h:1260-1263
        a["materials"] = mats
        a["lut_rgba"] = lut
        cam = dict(base.camera.as_dict())
        cam["pos"] = [float(cam["pos"][0] - rng.uniform(0, 2.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))]
        d = new_device()
        try:
            d.upload_arrays(a)
            d.set_transforms(a["transforms"], a["inv_transforms"])
            d.set_lut(lut)
            integ = [hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS, hjr.INTEGRATOR_PT][trial % 3]
            p = hjr.make_params(72, 48, 5, cam, frame=trial, seed=trial + 3, integrator=integ, sky=(0.8, 0.7, 0.6), ibl_intensity=0.9)
            color, albedo, normal = d.render(p)
        finally:
            d.close()
        op = ob.make_params(72, 48, 5, cam, frame=trial, seed=trial + 3, integrator=integ, sky=(0.8, 0.7, 0.6), ibl_intensity=0.9)
        oc, oa, on, st = ob.OracleScene(a, ob.MATH_PORTABLE).render(op)
        same_nan = np.isnan(color) == np.isnan(oc)
        assert same_nan.all()
        assert_bitexact(color, oc, "trial %d colour" % trial)
        assert_bitexact(albedo, oa, "trial %d albedo" % trial)


def test_no_lights_and_black_sky():
    """light_prim_count < 1: NEE adds nothing (the reference reads an uninitialised emission there); black sky."""
    base = Cornell()
    a = dict(base.arrays)
    mats = a["materials"].copy()
    mats[3]["is_light"] = 0
    mats[3]["emission"] = 0
    a["materials"] = mats
    a["light_prim_ids"] = np.zeros(0, np.uint32)
    a["light_prim_emission"] = np.zeros(0, np.float32)
    d = new_device()
    try:
        d.upload_arrays(a)
        d.set_transforms(a["transforms"], a["inv_transforms"])
        for integ in (hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS):
            p = base.hjr_params(64, 40, 3, integrator=integ, sky=(0.0, 0.0, 0.0))
            color, _, _ = d.render(p)
            oc, _, _, _ = ob.OracleScene(a, ob.MATH_PORTABLE).render(base.oracle_params(64, 40, 3, integrator=integ, sky=(0.0, 0.0, 0.0)))
            assert_bitexact(color, oc, "no lights")
            assert (color[..., :3] == 0).all()
    finally:
        d.close()


def test_normal_mapped_scene(tmp_path):
    """Material.normal_tex (gltfloader.h:1168-1175, bound at renderer.h:680): tangent-space normal map sampled in the closest-hit code
    (build-defined frame: per-triangle tangent from the uv deltas).  A generated wavy normal map on the textured box and on a wall:
    GPU == oracle bit for bit in both kernel families, and the map changes the picture."""
    from scene_util import make_normal_mapped_scene
    from test_gpu_variants import knobs
    s, n_mapped = make_normal_mapped_scene(tmp_path)
    assert s.scene.view.n_textures == 2 and sum(int(m["normal_tex"] >= 0) for m in s.arrays["materials"]) == n_mapped
    for pipe in ("mega", "wf"):
        with knobs(HJR_PIPELINE=pipe):
            with_map, _ = check(s, 112, 80, 6)
            check(s, 64, 48, 3, integrator=hjr.INTEGRATOR_MIS)
    plain, _ = check(Cornell("render_option_tex.json"), 112, 80, 6)
    assert not np.array_equal(with_map, plain)


@pytest.mark.parametrize("pipe", ["mega", "wf"])
def test_empty_scene_is_all_sky(pipe):
    """No triangles at all (the frame builder emits a single BVH4 root with four empty slots): every path misses, every pixel is
    scene_sky_default x IBL_intensity, in both kernel families and for every integrator."""
    from test_gpu_variants import knobs
    base = Cornell()
    a = dict(base.arrays)
    z = np.zeros(0, np.float32)
    a.update(vertices=z, normals=z, texcoords=z, indices=np.zeros(0, np.uint32), material_ids=np.zeros(0, np.uint32),
             prim_offsets=np.zeros(0, np.uint32), light_prim_ids=np.zeros(0, np.uint32), light_prim_emission=z)
    with knobs(HJR_PIPELINE=pipe):
        d = new_device()
        try:
            d.upload_arrays(a)
            d.set_transforms(np.zeros((0, 12), np.float32), np.zeros((0, 12), np.float32))
            for integ in (hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_PT, hjr.INTEGRATOR_MIS):
                color, albedo, normal = d.render(base.hjr_params(40, 24, 9, integrator=integ, sky=(0.25, 0.5, 0.75), ibl_intensity=2.0))
                assert d.stats()["pipeline"] == {"mega": 0, "wf": 1}[pipe] and d.stats()["n_triangles"] == 0
                exp = np.float32(9) * np.array([0.5, 1.0, 1.5], np.float32) * np.float32(1.0 / 9.0)  # nine equal samples summed, then x 1/spp
                assert np.allclose(color[..., :3], exp, rtol=1e-6) and (color[..., 3] == 1).all()
                assert (albedo[..., :3] == 0).all() and (normal[..., :3] == 0).all()
        finally:
            d.close()


def test_frame_from_the_product_loader_equals_oracle_on_an_independent_reading_of_the_gltf():
    """Every other GPU parity test hands the oracle the scene arrays of the PRODUCT's glTF loader, so a loader regression (a wrong vertex,
    normal, uv, material field, light list or instance offset) would change both sides alike and stay invisible.  Here the oracle gets the
    geometry, materials and light table from tests/gltf_ref.py — an independent numpy reading of the same .gltf / .bin following
    gltfloader.h:1068-1601 — and only the per-frame transforms and the camera from the product (the CPU tests check those against hand-derived
    values).  The GPU frame comes through the product loader as usual; the three AOVs must agree bit for bit."""
    import gltf_ref
    s = Cornell()
    ref = gltf_ref.load(os.path.join(hjr.ASSETS, "Model", "test_gltf"), "cornelbox.gltf")
    mats = np.zeros(len(ref["materials"]), dtype=ob.MATERIAL_DTYPE)
    for i, m in enumerate(ref["materials"]):
        for k in ("basecolor", "metallic", "roughness", "sheen", "clearcoat", "ior", "transmission", "emission", "is_light", "ideal_specular", "is_thinfilm"):
            mats[i][k] = m[k]
        for k in ("basecolor_tex", "metallic_roughness_tex", "normal_tex", "emission_tex"):
            mats[i][k] = -1  # cornelbox.gltf binds no texture
    n_tri = ref["material_ids"].size
    arrays = dict(vertices=ref["vertices"], normals=ref["normals"], texcoords=ref["texcoords"], indices=np.arange(3 * n_tri, dtype=np.uint32),
                  material_ids=ref["material_ids"], prim_offsets=ref["prim_offsets"], materials=mats, light_prim_ids=ref["light_prim_ids"],
                  light_prim_emission=ref["light_prim_emission"], transforms=s.arrays["transforms"], inv_transforms=s.arrays["inv_transforms"])
    d = s.device()
    try:
        w, h, spp = 112, 80, 6
        for integ in (hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS):
            color, albedo, normal = d.render(s.hjr_params(w, h, spp, integrator=integ))
            oc, oa, on, _ = ob.OracleScene(arrays, ob.MATH_PORTABLE).render(s.oracle_params(w, h, spp, integrator=integ))
            assert_bitexact(color, oc, "aov_color (integrator %d)" % integ)
            assert_bitexact(albedo, oa, "aov_albedo")
            assert_bitexact(normal, on, "aov_normal")
    finally:
        d.close()


def test_preview_buffer_on_the_device():
    """hjr_preview_device: the raygen's `uchar4 image` (renderer.h:1102, 1175) — the colour AOV through a tonemapper of kernel/color.h and the sRGB /
    quantise stage, on the device.  Against the host form of the same arithmetic (hjr_tonemap_to_srgb8, libm): equal, but for single pixels one
    code value off where the device's pow / exp rounds differently right at a quantisation boundary."""
    import torch
    s = Cornell()
    d = s.device()
    try:
        w, h = 200, 120
        fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        d.render_device(s.hjr_params(w, h, 16), fb.data_ptr())
        d.synchronize()
        host_in = fb.cpu().numpy()
        ramp = torch.linspace(0.0, 4.0, w * h, device="cuda").reshape(h, w)  # every code value and the shoulder of the tonemappers
        fb2 = torch.stack([ramp, ramp * 0.5, ramp * ramp, torch.ones_like(ramp)], dim=-1).contiguous()
        for img, name in ((fb, "frame"), (fb2, "ramp")):
            for mode in (hjr.TONEMAP_NONE, hjr.TONEMAP_UCHIMURA, hjr.TONEMAP_ACES):
                out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
                d.preview_device(img.data_ptr(), w, h, mode, out.data_ptr())
                d.synchronize()
                got = out.cpu().numpy()
                exp = hjr.tonemap_to_srgb8(img.cpu().numpy(), mode)
                diff = np.abs(got.astype(np.int32) - exp.astype(np.int32))
                assert diff.max() <= 1, (name, mode, int(diff.max()))
                assert (diff != 0).mean() < 1e-3, (name, mode, float((diff != 0).mean()))
                assert (got[..., 3] == 255).all()
        assert np.isfinite(host_in).all()
    finally:
        d.close()
