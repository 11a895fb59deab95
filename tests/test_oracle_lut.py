"""Thin-film LUT lookup semantics (disneyBRDF.h:11-14,213-217; sampler state renderer.h:854-898): wrap addressing,
texel-centre offset, bilinear weights, and the effect on the Disney specular F0."""
import ctypes as C

import numpy as np

import oracle_binding as ob
from scene_util import load_lut

L = ob.lib()


def fetch(lut, u, v):
    o = ob.F3()
    L.hjo_lut_fetch(lut.ctypes.data, lut.shape[1], lut.shape[0], u, v, o)
    return np.array(o, np.float32)


def test_texel_centres_wrap_and_bilinear():
    lut = np.zeros((4, 8, 4), np.uint8)
    lut[..., 0] = np.arange(8)[None, :] * 30
    lut[..., 1] = np.arange(4)[:, None] * 60
    lut[..., 3] = 255
    for i in range(8):
        for j in range(4):
            got = fetch(lut, (i + 0.5) / 8, (j + 0.5) / 4)  # texel centres return the texel (normalised float read)
            assert np.allclose(got[:2], [i * 30 / 255, j * 60 / 255], atol=1e-6)
    mid = fetch(lut, 1.0 / 8 + 0.5 / 8 + 0.5 / 8, 0.5 / 4)  # halfway between texel 1 and 2
    assert np.isclose(mid[0], 0.5 * (30 + 60) / 255, atol=1e-6)
    assert np.allclose(fetch(lut, 0.5 / 8 + 1.0, 0.5 / 4 - 2.0), fetch(lut, 0.5 / 8, 0.5 / 4))  # cudaAddressModeWrap
    edge = fetch(lut, 0.0, 0.5 / 4)  # u = 0 blends the last and first column
    assert np.isclose(edge[0], 0.5 * (0 + 210) / 255, atol=1e-6)


def test_thinfilm_changes_specular_f0():
    lut = load_lut()
    assert lut.shape == (256, 256, 4)
    m = ob.Material()
    m.basecolor = ob.F3(0.35, 0.8, 0.8)
    m.roughness, m.metallic, m.ior = 0.3, 0.0, 1.0
    wo = ob.F3(0.3, 0.9, 0.1)
    wi = ob.F3(-0.25, 0.93, 0.05)
    f0, f1, f2 = ob.F3(), ob.F3(), ob.F3()
    L.hjo_bsdf_eval(ob.MATH_PORTABLE, C.byref(m), wo, wi, None, 0, 0, f0)
    m.is_thinfilm = 1
    L.hjo_bsdf_eval(ob.MATH_PORTABLE, C.byref(m), wo, wi, lut.ctypes.data, 256, 256, f1)
    L.hjo_bsdf_eval(ob.MATH_PORTABLE, C.byref(m), wo, wi, None, 0, 0, f2)  # no LUT bound: F0 = 0
    a, b, c = np.array(f0), np.array(f1), np.array(f2)
    assert not np.allclose(a, b) and not np.allclose(b, c)
    assert len(set(np.round(b, 6))) > 1  # the film tints the highlight: channels differ although base colour y == z
