"""Every shipped kernel layout of both kernel families under a bit-exact parity test (VERDICT r01 task 1).

hjr_launch.hip.h::hjr_launch dispatches four layouts (hjr_stats.lds_mode): 0 = BVH4 read from memory, 1 = BVH2 staged in LDS with
32-bit stack entries, 2 = BVH2 in LDS with 16-bit stack entries, 3 = BVH2 read from memory; the memory-path layouts keep the
top of a lane's traversal stack in LDS and overflow into an HBM buffer.  The bundled scene only ever selects layout 1, so each
other layout is forced here (hjr_set_option: layout options act when the frame data is built, the others at the launch) and checked,
for NEE / Pathtrace / MIS with and without the albedo / normal AOVs, against the oracle's PORTABLE mode: same bar as
test_gpu_parity.py.  The layout actually used and the overflow activity are asserted through hjr_stats.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
from scene_util import Cornell, StressScene, device_options, hjr
from test_gpu_parity import assert_bitexact

pytestmark = pytest.mark.gpu

ALL_INTEGRATORS = (hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_PT, hjr.INTEGRATOR_MIS)


def knobs(**kv):
    """Options for the devices created inside the `with` block, spelled like the environment variables rounds 1 - 2 used
    (HJR_LDS_BVH=0 -> hjr_set_option("lds_bvh", 0)); the library itself reads no environment any more."""
    return device_options(**{k[4:].lower(): v for k, v in kv.items()})


_oracle_cache = {}


def oracle_frame(scene, key, w, h, spp, integrator):
    k = (key, w, h, spp, integrator)
    if k not in _oracle_cache:
        osc = ob.OracleScene(scene.arrays, ob.MATH_PORTABLE)
        oc, oa, on, st = osc.render(scene.oracle_params(w, h, spp, integrator=integrator))
        assert st["nan_samples"] == 0
        _oracle_cache[k] = (oc, oa, on)
    return _oracle_cache[k]


def check_layout(scene, key, env, expect_mode, w=96, h=64, spp=4, integrators=ALL_INTEGRATORS, expect_overflow=False, pipeline="mega"):
    """Renders with the knobs set, asserts the layout, compares all AOVs (full variant) and the colour-only (lean) variant."""
    env = dict(env)
    if pipeline is not None:  # None = let the library choose the kernel family (test_pipeline_selection)
        env["HJR_PIPELINE"] = pipeline
    with knobs(**env):
        d = scene.device()  # the options are set on the new context before its frame data is built
        try:
            for integ in integrators:
                oc, oa, on = oracle_frame(scene, key, w, h, spp, integ)
                color, albedo, normal = d.render(scene.hjr_params(w, h, spp, integrator=integ))
                st = d.stats()
                assert st["lds_mode"] == expect_mode, (st["lds_mode"], expect_mode)
                if pipeline is not None:
                    assert st["pipeline"] == {"mega": 0, "wf": 1}[pipeline], st
                assert_bitexact(color, oc, "aov_color (AOVS on, integrator %d)" % integ)
                assert_bitexact(albedo, oa, "aov_albedo")
                assert_bitexact(normal, on, "aov_normal")
                lean, _, _ = d.render(scene.hjr_params(w, h, spp, integrator=integ), want_aovs=False)
                assert_bitexact(lean, oc, "aov_color (AOVS off, integrator %d)" % integ)
                if expect_overflow:
                    assert st["stack_need"] > st["stack_lds_entries"] > 0, st
                    p = scene.hjr_params(w, h, spp, integrator=integ, flags=hjr.FLAG_STATS)
                    counted, _, _ = d.render(p, want_aovs=False)
                    assert_bitexact(counted, oc, "counting variant")
                    assert d.stats()["stack_overflow_pushes"] > 0, "the overflow branch of LaneStack::put never ran"
            return d.stats()
        finally:
            d.close()


@pytest.fixture(scope="module")
def cornell():
    return Cornell()


def test_default_layout_is_lds32(cornell):
    st = check_layout(cornell, "cornell", {}, expect_mode=1)
    assert st["stack_need"] == st["bvh_depth"] + 2


def test_lds_bvh2_with_16bit_stack_forced_on_cornell(cornell):
    """lds_mode 2 (STACK16): the 16-bit entries are normally chosen only when 32-bit ones do not fit; HJR_LDS_STACK16=1 prefers them."""
    check_layout(cornell, "cornell", {"HJR_LDS_STACK16": 1}, expect_mode=2)


def test_lds_bvh2_with_16bit_stack_chosen_by_the_builder(tmp_path):
    """A generated scene inside the window (~1.2-1.5 k triangles) where the builder itself selects the 16-bit stack layout."""
    s = StressScene(tmp_path, spheres=6, segments=16)
    assert s.scene.view.n_triangles == 12 + 6 * 224
    check_layout(s, "ss6x16", {}, expect_mode=2, w=80, h=45, spp=3)


def test_bvh4_from_memory_on_cornell(cornell):
    """lds_mode 0 on the bundled scene (HJR_LDS_BVH=0), whole stack in LDS (14 entries < 16)."""
    st = check_layout(cornell, "cornell", {"HJR_LDS_BVH": 0}, expect_mode=0)
    assert st["stack_lds_entries"] == min(st["stack_need"], 16)


def test_bvh2_from_memory_on_cornell(cornell):
    """lds_mode 3: BVH2 nodes read from memory (HJR_BVH_WIDTH=2 with the LDS staging off)."""
    check_layout(cornell, "cornell", {"HJR_LDS_BVH": 0, "HJR_BVH_WIDTH": 2}, expect_mode=3)


@pytest.mark.parametrize("width,mode", [(4, 0), (2, 3)])
def test_stack_overflow_path(cornell, width, mode):
    """Two LDS entries per lane: every deeper push goes through the HBM overflow buffer ([level][lane]) and comes back."""
    check_layout(cornell, "cornell", {"HJR_LDS_BVH": 0, "HJR_BVH_WIDTH": width, "HJR_SHORT_STACK": 2}, expect_mode=mode,
                 expect_overflow=True)


def test_memory_layouts_on_a_deep_scene(tmp_path):
    """~60 k triangles (BVH4 stack need > 16): default short stack of 16 with real overflow, and the BVH2-from-memory layout."""
    s = StressScene(tmp_path, spheres=8, segments=88)
    st = check_layout(s, "ss8x88", {}, expect_mode=0, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE,))
    assert st["stack_need"] > 16 and st["stack_lds_entries"] == 16
    check_layout(s, "ss8x88", {"HJR_BVH_WIDTH": 2}, expect_mode=3, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS))
    check_layout(s, "ss8x88", {"HJR_SHORT_STACK": 3}, expect_mode=0, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE,),
                 expect_overflow=True)


@pytest.mark.parametrize("passes", [0, 1, 8])
def test_bvh_refinement_changes_the_tree_not_the_frame(cornell, tmp_path, passes):
    """Option "bvh_refine" (insertion-based refinement of the built BVH2, host/frame.cpp::Refine): none / one / eight passes give a
    different tree (depth) and the same bits, in the LDS layout of the bundled scene and the memory layouts of a 60 k-triangle one."""
    st = check_layout(cornell, "cornell", {"HJR_BVH_REFINE": passes}, expect_mode=1, integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS))
    assert st["bvh_depth"] == (12 if passes == 0 else st["bvh_depth"]) and (passes == 0 or st["bvh_depth"] != 12), st
    s = StressScene(tmp_path, spheres=8, segments=88)
    check_layout(s, "ss8x88", {"HJR_BVH_REFINE": passes}, expect_mode=0, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE,))
    check_layout(s, "ss8x88", {"HJR_BVH_REFINE": passes, "HJR_BVH_WIDTH": 2}, expect_mode=3, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE,))


# ---- the workgroup-local wavefront kernel family (hjr_wavefront.hip.h): same layouts, same bar
WF = "wf"


def test_wavefront_default_layout(cornell):
    check_layout(cornell, "cornell", {}, expect_mode=1, pipeline=WF)


def test_wavefront_sizes_and_chunks(cornell):
    """Ragged frames (items outside the frame edge are skipped and retried), one-pixel frames, several sample chunks per pixel,
    small context pools (more turnover per context), and a pool larger than the frame."""
    for (w, h, spp) in [(1, 1, 1), (7, 5, 3), (40, 24, 40), (200, 120, 35)]:
        check_layout(cornell, "cornell", {}, expect_mode=1, w=w, h=h, spp=spp, integrators=(hjr.INTEGRATOR_NEE,), pipeline=WF)
    check_layout(cornell, "cornell", {"HJR_WF_CAP": 256}, expect_mode=1, w=64, h=48, spp=9, pipeline=WF)
    check_layout(cornell, "cornell", {"HJR_WF_CAP": 64}, expect_mode=1, w=33, h=17, spp=5, integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS), pipeline=WF)


def test_wavefront_lds16(cornell):
    check_layout(cornell, "cornell", {"HJR_LDS_STACK16": 1}, expect_mode=2, pipeline=WF)


@pytest.mark.parametrize("width,mode", [(4, 0), (2, 3)])
def test_wavefront_memory_layouts_and_overflow(cornell, width, mode):
    check_layout(cornell, "cornell", {"HJR_LDS_BVH": 0, "HJR_BVH_WIDTH": width}, expect_mode=mode, pipeline=WF)
    check_layout(cornell, "cornell", {"HJR_LDS_BVH": 0, "HJR_BVH_WIDTH": width, "HJR_SHORT_STACK": 2}, expect_mode=mode,
                 integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS), expect_overflow=True, pipeline=WF)


def test_wavefront_deep_scene(tmp_path):
    s = StressScene(tmp_path, spheres=8, segments=88)
    st = check_layout(s, "ss8x88", {}, expect_mode=0, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_PT), pipeline=WF)
    assert st["stack_need"] > 16 and st["stack_lds_entries"] == 16


@pytest.mark.parametrize("pipe", ["mega", WF])
@pytest.mark.parametrize("node_min", [1, 64])
def test_descent_threshold_extremes(cornell, tmp_path, pipe, node_min):
    """HJR_NODE_MIN (hjr_traverse.hip.h / wf_trace_stage): with 1 the lanes of a wave descend until every lane holds a leaf, with 64 a
    pass ends after a single node step unless all 64 lanes are still descending, so practically every lane carries an inner node from
    one pass to the next.  Same bits in both kernel families, for the LDS-resident layout, both memory layouts, the stack overflow
    path and a deep tree."""
    env = {"HJR_NODE_MIN": node_min}
    check_layout(cornell, "cornell", dict(env), expect_mode=1, w=64, h=48, spp=3, pipeline=pipe)
    check_layout(cornell, "cornell", dict(env, HJR_LDS_BVH=0), expect_mode=0, w=64, h=48, spp=3, pipeline=pipe)
    check_layout(cornell, "cornell", dict(env, HJR_LDS_BVH=0, HJR_BVH_WIDTH=2, HJR_SHORT_STACK=2), expect_mode=3, w=48, h=32, spp=2,
                 integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS), expect_overflow=True, pipeline=pipe)
    s = StressScene(tmp_path, spheres=8, segments=88)
    check_layout(s, "ss8x88", dict(env), expect_mode=0, w=64, h=36, spp=2, integrators=(hjr.INTEGRATOR_NEE,), pipeline=pipe)


@pytest.mark.parametrize("hold_min,hold_age", [(0, 1), (2, 1), (64, 9)])
def test_rare_class_hold_back(cornell, tmp_path, hold_min, hold_age):
    """HJR_HOLD_MIN / HJR_HOLD_AGE (megakernel, hjr_kernel.hip.h): hits on metallic (multiple-scattering GGX) surfaces wait until a wave
    holds hold_min of them or one has waited hold_age rounds.  Off, eager (2 / 1) and as lazy as it gets (64 / 9: every such hit waits
    nine rounds unless nothing else is left): same bits in the LDS-resident layouts (the kernels of the memory layouts do not hold)."""
    env = {"HJR_HOLD_MIN": hold_min, "HJR_HOLD_AGE": hold_age}
    check_layout(cornell, "cornell", dict(env), expect_mode=1, w=80, h=56, spp=5)
    check_layout(cornell, "cornell", dict(env, HJR_LDS_STACK16=1), expect_mode=2, w=64, h=48, spp=3, integrators=(hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS))
    s = Cornell("render_option_c2_nodiel.json")  # the sphere itself is metallic here: most hits of its tiles are of the held class
    check_layout(s, "cornell_nodiel", dict(env), expect_mode=1, w=80, h=56, spp=4, integrators=(hjr.INTEGRATOR_NEE,))


def test_wavefront_statistics_match_the_megakernel(cornell):
    """Ray / hit / sample counters are properties of the sample streams: both kernel families must report the same numbers."""
    res = {}
    for pipe in ("mega", WF):
        with knobs(HJR_PIPELINE=pipe):
            d = cornell.device()
            try:
                d.render(cornell.hjr_params(64, 64, 4, flags=hjr.FLAG_STATS), want_aovs=False)
                res[pipe] = d.stats()
            finally:
                d.close()
    for k in ("samples", "closest_rays", "shadow_rays", "shaded_hits", "light_samples", "nan_samples", "box_tests_closest",
              "tri_tests_closest", "box_tests_shadow", "tri_tests_shadow"):  # (LDS layout: both families walk the tree in the same order)
        assert res["mega"][k] == res[WF][k], (k, res["mega"][k], res[WF][k])


def test_full_size_frames_agree_between_the_families(cornell):
    """The bench size (1920x1080; 48 spp = six sample runs per pixel): the megakernel's and the wavefront kernel's frames are the same bits
    for NEE and MIS.  (The first queue protocol of the wavefront kernel was correct on small frames and lost entries from ~30 M samples on.)"""
    for integ in (hjr.INTEGRATOR_NEE, hjr.INTEGRATOR_MIS):
        frames = {}
        for pipe in ("mega", WF):
            with knobs(HJR_PIPELINE=pipe):
                d = cornell.device()
                try:
                    frames[pipe], _, _ = d.render(cornell.hjr_params(1920, 1080, 48, integrator=integ), want_aovs=False)
                    assert d.stats()["pipeline"] == {"mega": 0, WF: 1}[pipe]
                finally:
                    d.close()
        assert_bitexact(frames["mega"], frames[WF], "integrator %d at 1920x1080x48" % integ)


def test_pipeline_selection(cornell):
    """Without HJR_PIPELINE the library picks the kernel family per launch (hjr_launch.hip.h::hjr_launch): the wavefront kernels for MIS
    (any layout), the megakernel otherwise.  Whatever it picks, the bits are the oracle's."""
    if True:
        d = cornell.device()
        try:
            def pipe(integ, aovs):
                d.render(cornell.hjr_params(48, 32, 2, integrator=integ), want_aovs=aovs)
                return d.stats()["pipeline"]
            assert pipe(hjr.INTEGRATOR_NEE, False) == 0 and pipe(hjr.INTEGRATOR_NEE, True) == 0
            assert pipe(hjr.INTEGRATOR_MIS, False) == 1 and pipe(hjr.INTEGRATOR_MIS, True) == 1
            assert pipe(hjr.INTEGRATOR_PT, False) == 0
        finally:
            d.close()
        check_layout(cornell, "cornell", {}, expect_mode=1, pipeline=None)
        with knobs(HJR_LDS_BVH=0):
            d = cornell.device()
            try:
                d.render(cornell.hjr_params(48, 32, 2, integrator=hjr.INTEGRATOR_NEE), want_aovs=False)
                assert d.stats()["pipeline"] == 0  # NEE on a scene read from memory stays on the megakernel ...
                d.render(cornell.hjr_params(48, 32, 2, integrator=hjr.INTEGRATOR_MIS), want_aovs=False)
                assert d.stats()["pipeline"] == 1  # ... MIS is faster on the wavefront kernels in every layout
            finally:
                d.close()


def test_option_api_validates_keys_and_ranges(cornell):
    d = cornell.device()
    try:
        assert d.get_option("pipeline") == -1 and d.get_option("node_min") == -1  # defaults
        d.set_option("node_min", 7)
        assert d.get_option("node_min") == 7
        d.set_option("node_min", -1)
        assert d.get_option("node_min") == -1
        for key, bad in (("no_such_option", 1), ("node_min", 0), ("node_min", 65), ("bvh_width", 3), ("wf_cap", 100), ("pipeline", 3), ("leaf_max", 5)):
            with pytest.raises(hjr.HjrError):
                d.set_option(key, bad)
        with pytest.raises(hjr.HjrError):
            d.get_option("no_such_option")
        # a layout option acts at the next set_transforms: the same context goes from the LDS layout to BVH4 from memory and back
        p = cornell.hjr_params(48, 32, 2)
        a, _, _ = d.render(p, want_aovs=False)
        assert d.stats()["lds_mode"] == 1
        d.set_option("lds_bvh", 0)
        d.set_transforms(cornell.arrays["transforms"], cornell.arrays["inv_transforms"])
        b, _, _ = d.render(p, want_aovs=False)
        assert d.stats()["lds_mode"] == 0
        assert_bitexact(a, b, "LDS layout vs memory layout on one context")
        d.set_option("lds_bvh", -1)
        d.set_transforms(cornell.arrays["transforms"], cornell.arrays["inv_transforms"])
        d.render(p, want_aovs=False)
        assert d.stats()["lds_mode"] == 1
    finally:
        d.close()
