"""Pins oracle/ against every reference-derived known-answer value available.

Integer functions: exact.  Float functions: bit-exact in math mode HOSTF64 (the arithmetic the values
were produced with); <= 4 ulp in LIBM/PORTABLE modes (float-overload binding as under nvcc).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_binding as ob

f32 = np.float32
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_8c_kat.json")))
L = ob.lib()


def f32eq(a, b):
    """bit-exact float32 comparison of a computed value against a 9-significant-digit decimal."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=np.float64).astype(f32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b)


def ulps(a, b):
    a = np.asarray(a, dtype=f32).astype(np.float64)
    b = np.asarray(b, dtype=np.float64)
    sp = np.spacing(np.maximum(np.abs(b), 1e-30).astype(f32)).astype(np.float64)
    return np.max(np.abs(a - b) / sp)


def normalize(v):
    """sutil normalize: v * (1/sqrt(dot))"""
    v = np.asarray(v, dtype=f32)
    d = f32(f32(v[0] * v[0]) + f32(v[1] * v[1]))
    d = f32(d + f32(v[2] * v[2]))
    return (v * f32(f32(1) / np.sqrt(d))).astype(f32)


def state5(s):
    return (C.c_uint32 * 5)(s["n_spp"] & 0xFFFFFFFF, s["n_spp"] >> 32, s["scramble"], s["depth"], s["image_idx"])


def material(m=None, **kw):
    mt = ob.Material()
    mt.ior = 1.0
    d = dict(m or {})
    d.update(kw)
    for k, v in d.items():
        if k == "basecolor":
            mt.basecolor = ob.F3(*v)
        else:
            setattr(mt, k, v)
    return mt


def test_xxhash32_exact():
    for c in KAT["xxhash32_u4"]:
        assert L.hjo_xxhash32_u4(*c["in"]) == c["out"]


def test_cmj_permute_exact():
    for c in KAT["cmj_permute"]:
        assert L.hjo_cmj_permute(*c["in"]) == c["out"]


@pytest.fixture(scope="module")
def narrow(tmp_path_factory):
    """henjou-renderer_amd/csrc/hjr_cmj.h (the product's mod-2^k permutes and split xxhash32) compiled for the host."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("cmj") / "cmj_narrow.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(root, "tests", "native", "cmj_narrow_test.cpp"), "-o", so])
    N = C.CDLL(so)
    for f in ("sweep", "sweep_bijective"):
        getattr(N, f).restype = C.c_uint64
        getattr(N, f).argtypes = [C.c_uint64, C.c_uint64]
    for f, n in (("narrow_permute16", 2), ("narrow_permute4", 2), ("split_xxhash", 4)):
        getattr(N, f).restype = C.c_uint32
        getattr(N, f).argtypes = [C.c_uint32] * n
    return N


def test_product_cmj_integer_functions_equal_the_32_bit_loop(narrow):
    """Every i < l for 4 M random and 4096 structured p (l = 16 and l = 4), and the split hash, against the spelled-out originals."""
    assert narrow.sweep(4_000_000, 20260315) == 0
    assert narrow.sweep_bijective(500_000, 7) == 0


def test_product_cmj_integer_functions_equal_the_oracle_and_the_kat(narrow):
    for c in KAT["cmj_permute"]:
        i, l, p = c["in"]
        if l == 16 and i < 16:
            assert narrow.narrow_permute16(i, p) == c["out"]
        if l == 4 and i < 4:
            assert narrow.narrow_permute4(i, p) == c["out"]
    for c in KAT["xxhash32_u4"]:
        assert narrow.split_xxhash(*c["in"]) == c["out"]
    rng = np.random.default_rng(5)
    for p in rng.integers(0, 2**32, 3000, dtype=np.uint64):
        p = int(p)
        for i in range(16):
            assert narrow.narrow_permute16(i, p) == L.hjo_cmj_permute(i, 16, p)
        for i in range(4):
            assert narrow.narrow_permute4(i, p) == L.hjo_cmj_permute(i, 4, p)
    for a, b, c_, d in rng.integers(0, 2**32, (3000, 4), dtype=np.uint64):
        assert narrow.split_xxhash(int(a), int(b), int(c_), int(d)) == L.hjo_xxhash32_u4(int(a), int(b), int(c_), int(d))


def test_cmj_randfloat_and_cmj_bitexact():
    for c in KAT["cmj_randfloat"]:
        assert f32eq(L.hjo_cmj_randfloat(*c["in"]), c["out"])
    for c in KAT["cmj"]:
        o = (C.c_float * 2)()
        L.hjo_cmj(c["in"][0], c["in"][1], o)
        assert f32eq(list(o), c["out"])


def test_cmj_2d_sequences_bitexact():
    for seq in KAT["cmj_2d_sequences"]:
        st = state5(seq["state"])
        o = (C.c_float * 2)()
        for k, exp in enumerate(seq["out"]):
            L.hjo_cmj_2d(st, o)
            assert f32eq(list(o), exp), (k, list(o), exp)
            assert st[3] == seq["state"]["depth"] + k + 1


@pytest.mark.parametrize("mode,tol", [(ob.MATH_HOSTF64, 0), (ob.MATH_LIBM, 4), (ob.MATH_PORTABLE, 4)])
def test_math_helpers(mode, tol):
    c = KAT["cosineSampling"][0]
    w, p = ob.F3(), C.c_float()
    L.hjo_cosine_sampling(mode, c["in"][0], c["in"][1], w, C.byref(p))
    if tol == 0:
        assert f32eq(list(w), c["wi"]) and f32eq(p.value, c["pdf"])
    else:
        # wi.x is cos(3*pi/2)*0.5 ~ 6e-9: compare absolutely
        assert abs(w[0] - c["wi"][0]) < 1e-9 and ulps(list(w)[1:], c["wi"][1:]) <= tol and ulps(p.value, c["pdf"]) <= tol
    c = KAT["orthonormal_basis"][0]
    t, b = ob.F3(), ob.F3()
    L.hjo_orthonormal_basis(ob.F3(*normalize(c["n_unnormalized"])), t, b)
    assert f32eq(list(t), c["t"]) and f32eq(list(b), c["b"])
    c = KAT["refract"][0]
    r = ob.F3()
    assert L.hjo_refract(ob.F3(*normalize(c["v_unnormalized"])), ob.F3(*c["n"]), c["ior1"], c["ior2"], r) == 1
    assert f32eq(list(r), c["r"])
    c = KAT["shlickFresnel_ior"][0]
    wv = normalize([0.6, c["cos"], 0.0])
    got = L.hjo_schlick_ior(mode, c["no"], c["ni"], ob.F3(*wv), ob.F3(0, 1, 0))
    assert ulps(got, c["out"]) <= max(tol, 0) if tol else f32eq(got, c["out"])


@pytest.mark.parametrize("mode,tol", [(ob.MATH_HOSTF64, 0), (ob.MATH_LIBM, 4), (ob.MATH_PORTABLE, 8)])
def test_bsdf_kats(mode, tol):
    def check(got, exp):
        if tol == 0:
            assert f32eq(got, exp), (got, exp)
        else:
            assert ulps(got, exp) <= tol, (got, exp, ulps(got, exp))

    # Disney sample (disneyBRDF.h:237-307) then glass (BSDFs.h:419-469) on the continuing stream
    c = KAT["disney_sample"]
    m = material(c["material"])
    st = state5(c["state"])
    wo = ob.F3(*normalize(c["wo_unnormalized"]))
    f, wi, pdf = ob.F3(), ob.F3(0, 1, 0), C.c_float(1)
    L.hjo_bsdf_sample(mode, C.byref(m), 0, wo, st, f, wi, C.byref(pdf))
    check(list(f), c["f"]); check(pdf.value, c["pdf"]); check(list(wi), c["wi"])
    assert st[3] == KAT["glass_sample"]["state"]["depth"]
    g = KAT["glass_sample"]
    m.ior = g["ior"]
    L.hjo_bsdf_sample(mode, C.byref(m), 1, wo, st, f, wi, C.byref(pdf))
    check(list(f), g["f"]); check(list(wi), g["wi"])
    assert pdf.value == 1.0

    # multiple-scattering GGX random walk (BSDFs.h:784-851)
    c = KAT["msggx_sample"]
    m = material(c["material"])
    st = state5(c["state"])
    f, wi, pdf = ob.F3(), ob.F3(0, 1, 0), C.c_float(1)
    L.hjo_bsdf_sample(mode, C.byref(m), 2, ob.F3(*normalize(c["wo_unnormalized"])), st, f, wi, C.byref(pdf))
    check(list(f), c["weight"]); check(pdf.value, c["pdf"]); check(list(wi), c["wi"])
    assert st[3] == c["depth_after"]

    # Disney evaluate + getPDF (disneyBRDF.h:179-235, 309-326)
    c = KAT["disney_eval"]
    m = material(c["material"])
    wo = ob.F3(*normalize(c["wo_unnormalized"]))
    wi = ob.F3(*normalize(c["wi_unnormalized"]))
    L.hjo_bsdf_eval(mode, C.byref(m), wo, wi, None, 0, 0, f)
    check(list(f), c["f"])
    check(L.hjo_bsdf_pdf(mode, C.byref(m), wo, wi), c["pdf"])


def test_portable_math_close_to_libm():
    rng = np.random.default_rng(7)
    xs = rng.uniform(0, 2 * np.pi, 20000).astype(f32)
    s = np.array([L.hjo_p_sin(float(x)) for x in xs[:4000]], dtype=f32)
    c = np.array([L.hjo_p_cos(float(x)) for x in xs[:4000]], dtype=f32)
    assert np.max(np.abs(s - np.sin(xs[:4000].astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(c - np.cos(xs[:4000].astype(np.float64)))) < 2.5e-7
    us = rng.uniform(-1, 1, 4000).astype(f32)
    a = np.array([L.hjo_p_acos(float(x)) for x in us], dtype=f32)
    assert np.max(np.abs(a - np.arccos(us.astype(np.float64)))) < 6e-7
    assert L.hjo_p_acos(1.0) == 0.0 and abs(L.hjo_p_acos(-1.0) - np.pi) < 5e-7
    bx = rng.uniform(1e-4, 1.0, 4000).astype(f32)
    by = rng.uniform(0.0, 30.0, 4000).astype(f32)
    p = np.array([L.hjo_p_pow(float(x), float(y)) for x, y in zip(bx, by)], dtype=np.float64)
    ref = np.power(bx.astype(np.float64), by.astype(np.float64))
    ok = ref > 1e-30
    assert np.max(np.abs(p[ok] - ref[ok]) / ref[ok]) < 3e-5
    # special cases the random walk relies on (BSDFs.h:561,583)
    assert L.hjo_p_pow(0.0, 0.0) == 1.0 and L.hjo_p_pow(0.5, 0.0) == 1.0 and L.hjo_p_pow(1.0, 1e30) == 1.0
    assert L.hjo_p_pow(0.0, 2.0) == 0.0 and L.hjo_p_pow(0.5, float("inf")) == 0.0
    assert L.hjo_p_pow(0.5, float("-inf")) == float("inf")
    assert L.hjo_p_pow5(-0.5) == -0.03125


def test_srgb_output_stage():
    # toSRGB + quantizeUnsignedChar (renderer.h:73-101)
    px = np.array([[0.0, 0.0031308, 0.5, 1.0], [1.0, 2.0, 0.002, 1.0], [0.18, 0.999, 1e-5, 0.0]], dtype=f32)
    out = np.zeros((3, 4), dtype=np.uint8)
    L.hjo_float4_to_srgb8(px.ctypes.data, out.ctypes.data, 3)

    def ref(c):
        c = f32(c)
        s = f32(12.92) * c if c < f32(0.0031308) else f32(f32(1.055) * f32(np.float64(c) ** (1 / 2.4)) - f32(0.055))
        return min(int(f32(s * f32(256.0))), 255)
    for i in range(3):
        for k in range(3):
            assert abs(int(out[i, k]) - ref(px[i, k])) <= 1
        assert out[i, 3] == 255
    assert out[0, 0] == 0 and out[1, 0] == 255 and out[1, 1] == 255
