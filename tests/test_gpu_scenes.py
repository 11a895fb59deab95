"""GPU parity on the other BASELINE.json configurations: thin-film LUT BSDF (configs[2]), negative-IOR glass with
ior 1.5 (configs[3]), the diffuse/specular-only variant (configs[1] strict reading) and the synthetic many-triangle
stress scene.  Same bar as test_gpu_parity.py: bit-exact against the oracle's PORTABLE mode."""
import numpy as np
import pytest

import oracle_binding as ob
from scene_util import Cornell, StressScene, hjr, load_lut
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
