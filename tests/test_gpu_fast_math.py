"""HJR_FLAG_FAST_MATH: the opt-in approximate-arithmetic kernels (hardware reciprocal / square root / sine / cosine / power and fused
multiply-adds in the shading code — what the reference's own nvcc --use_fast_math build does).  They are NOT bit-exact; the bar is the
metric's own tolerance: per-pixel RMSE < 1e-3 at 1024 spp at equal sample streams (BASELINE.json north_star), written here."""
import numpy as np
import pytest

import oracle_binding as ob
from scene_util import Cornell, device_options, hjr, load_lut

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: per-pixel RMSE < 1e-3 at 1024 spp


def rmse(a, b):
    return float(np.sqrt(np.mean((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2)))


@pytest.mark.parametrize("config", ["render_option_c2.json", "render_option_c3.json", "render_option_c4.json"])
def test_fast_math_rmse_1024spp_vs_libm_oracle(config):
    """configs[1..3] (plain, thin-film LUT, ior-1.5 negative-index glass) at 256x256x1024 against the CPU restatement with glibc
    transcendentals and IEEE division, identical sample streams."""
    s = Cornell(config)
    w, h, spp = 256, 256, 1024
    arrays = dict(s.arrays)
    d = s.device()
    try:
        if config == "render_option_c3.json":
            lut = load_lut()
            d.set_lut(lut)
            arrays["lut_rgba"] = lut
        fast, _, _ = d.render(s.hjr_params(w, h, spp, flags=hjr.FLAG_FAST_MATH), want_aovs=False)
        assert d.stats()["fast_math"] == 1 and d.stats()["pipeline"] == 0
        exact, _, _ = d.render(s.hjr_params(w, h, spp), want_aovs=False)
        assert d.stats()["fast_math"] == 0
    finally:
        d.close()
    assert np.isfinite(fast).all() and (fast[..., 3] == 1).all()
    osc = ob.OracleScene(arrays, ob.MATH_LIBM)
    oc, _, _, _ = osc.render(s.oracle_params(w, h, spp), want_aovs=False)
    r_oracle, r_exact = rmse(fast, oc), rmse(fast, exact)
    print("fast-math RMSE %s 256x256x1024: %.3e vs LIBM oracle, %.3e vs the exact kernel" % (config, r_oracle, r_exact))
    assert r_oracle < TOL and r_exact < TOL
    assert not np.array_equal(fast, exact)  # (it IS a different arithmetic; equality would mean the flag did nothing)


@pytest.mark.parametrize("integrator", [hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_PT, hjr.INTEGRATOR_MIS])
def test_fast_math_full_size_vs_exact_kernel(integrator):
    """The bench size: 1920x1080x256 against the bit-exact kernel of the same library; the albedo / normal AOVs are first-hit quantities of an
    unchanged traversal, so they differ at most by the jitter's rounding.  MIS ignores the flag (its exact wavefront kernels are faster than an
    approximate megakernel): the same bits come back and hjr_stats.fast_math says 0."""
    s = Cornell("render_option_c2.json")
    d = s.device()
    try:
        w, h, spp = 1920, 1080, 256
        fast, fa, fn = d.render(s.hjr_params(w, h, spp, integrator=integrator, flags=hjr.FLAG_FAST_MATH))
        ran_fast = d.stats()["fast_math"]
        exact, ea, en = d.render(s.hjr_params(w, h, spp, integrator=integrator))
    finally:
        d.close()
    assert ran_fast == (0 if integrator == hjr.INTEGRATOR_MIS else 1)
    if integrator == hjr.INTEGRATOR_MIS:
        assert np.array_equal(fast, exact)
        return
    r = rmse(fast, exact)
    print("fast-math RMSE 1920x1080x256 integrator %d vs exact kernel: %.3e (albedo %.3e, normal %.3e)" % (integrator, r, rmse(fa, ea), rmse(fn, en)))
    assert r < 4 * TOL  # 256 spp: the 1024-spp tolerance scaled by sqrt(1024 / 256) = 2, with margin
    assert rmse(fa, ea) < TOL and rmse(fn, en) < TOL


def test_fast_math_every_layout_and_counting_launch():
    """The flag reaches every megakernel layout (LDS 32-bit / 16-bit stacks, BVH4 / BVH2 from memory) and a counting launch ignores it."""
    s = Cornell()
    ref = None
    for opts, mode in (({}, 1), ({"lds_stack16": 1}, 2), ({"lds_bvh": 0}, 0), ({"lds_bvh": 0, "bvh_width": 2}, 3)):
        with device_options(**opts):
            d = s.device()
            try:
                c, _, _ = d.render(s.hjr_params(96, 64, 8, flags=hjr.FLAG_FAST_MATH), want_aovs=False)
                st = d.stats()
                assert st["lds_mode"] == mode and st["fast_math"] == 1
                if ref is None:
                    ref = c
                    e, _, _ = d.render(s.hjr_params(96, 64, 8), want_aovs=False)
                    assert rmse(c, e) < 0.05  # 8 spp: only a sanity bound
                    k, _, _ = d.render(s.hjr_params(96, 64, 8, flags=hjr.FLAG_FAST_MATH | hjr.FLAG_STATS), want_aovs=False)
                    assert d.stats()["fast_math"] == 0 and np.array_equal(k, e)  # counting launches stay exact
                else:  # the shading arithmetic is the same in every layout; which hit wins never depends on the layout
                    assert np.array_equal(c, ref), "fast-math frames differ between layouts %s" % (opts,)
            finally:
                d.close()
