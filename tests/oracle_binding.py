"""ctypes binding of oracle/libhjr_oracle.so — the CPU oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libhjr_oracle.so")

MATH_LIBM, MATH_PORTABLE, MATH_HOSTF64 = 0, 1, 2
INTEGRATOR_NEE, INTEGRATOR_PT, INTEGRATOR_MIS = 0, 1, 2

F3 = C.c_float * 3


class Material(C.Structure):
    _fields_ = [("basecolor", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("sheen", C.c_float), ("clearcoat", C.c_float), ("ior", C.c_float),
                ("transmission", C.c_float), ("emission", C.c_float * 3), ("is_light", C.c_int32),
                ("ideal_specular", C.c_int32), ("is_thinfilm", C.c_int32), ("basecolor_tex", C.c_int32),
                ("metallic_roughness_tex", C.c_int32), ("normal_tex", C.c_int32), ("emission_tex", C.c_int32),
                ("_reserved", C.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.basecolor_tex = self.metallic_roughness_tex = self.normal_tex = self.emission_tex = -1


MATERIAL_DTYPE = np.dtype([("basecolor", "<f4", 3), ("metallic", "<f4"), ("roughness", "<f4"),
                           ("sheen", "<f4"), ("clearcoat", "<f4"), ("ior", "<f4"),
                           ("transmission", "<f4"), ("emission", "<f4", 3), ("is_light", "<i4"),
                           ("ideal_specular", "<i4"), ("is_thinfilm", "<i4"), ("basecolor_tex", "<i4"),
                           ("metallic_roughness_tex", "<i4"), ("normal_tex", "<i4"), ("emission_tex", "<i4"),
                           ("_reserved", "<i4")])
assert MATERIAL_DTYPE.itemsize == 80 and C.sizeof(Material) == 80


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("srgb", C.c_int32),
                ("_reserved", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_instances", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_lights", C.c_uint32), ("vertices", C.c_void_p), ("normals", C.c_void_p),
                ("texcoords", C.c_void_p), ("indices", C.c_void_p), ("material_ids", C.c_void_p),
                ("prim_offsets", C.c_void_p), ("transforms", C.c_void_p), ("inv_transforms", C.c_void_p),
                ("materials", C.c_void_p), ("light_prim_ids", C.c_void_p),
                ("light_prim_emission", C.c_void_p), ("lut_rgba", C.c_void_p),
                ("lut_w", C.c_int32), ("lut_h", C.c_int32), ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                ("sky_w", C.c_int32), ("sky_h", C.c_int32), ("sky_rgba", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("frame", C.c_uint32),
                ("seed", C.c_uint32), ("integrator", C.c_uint32), ("cam_pos", C.c_float * 3),
                ("cam_dir", C.c_float * 3), ("cam_up", C.c_float * 3), ("cam_right", C.c_float * 3),
                ("cam_f", C.c_float), ("sky", C.c_float * 3), ("ibl_intensity", C.c_float),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "box_tests_closest",
                                           "tri_tests_closest", "box_tests_shadow", "tri_tests_shadow",
                                           "shaded_hits", "light_samples", "nan_samples")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.hjo_xxhash32_u4.restype = C.c_uint32
        L.hjo_xxhash32_u4.argtypes = [C.c_uint32] * 4
        L.hjo_cmj_permute.restype = C.c_uint32
        L.hjo_cmj_permute.argtypes = [C.c_uint32] * 3
        L.hjo_cmj_randfloat.restype = C.c_float
        L.hjo_cmj_randfloat.argtypes = [C.c_uint32] * 2
        L.hjo_cmj.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.hjo_cmj_2d.argtypes = [C.c_void_p, C.c_void_p]
        L.hjo_cosine_sampling.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.hjo_orthonormal_basis.argtypes = [C.c_void_p] * 3
        L.hjo_refract.restype = C.c_int
        L.hjo_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.hjo_schlick_ior.restype = C.c_float
        L.hjo_schlick_ior.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.hjo_bsdf_sample.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.hjo_bsdf_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p]
        L.hjo_bsdf_pdf.restype = C.c_float
        L.hjo_bsdf_pdf.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hjo_p_atan2.restype = C.c_float
        L.hjo_p_atan2.argtypes = [C.c_float, C.c_float]
        L.hjo_tex_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.hjo_sky_fetch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for n in ("hjo_p_sin", "hjo_p_cos", "hjo_p_acos", "hjo_p_pow5"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.hjo_p_pow.restype = C.c_float
        L.hjo_p_pow.argtypes = [C.c_float, C.c_float]
        L.hjo_lut_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.hjo_float4_to_srgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.hjo_tonemap_to_srgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.hjo_denoise.restype = C.c_int
        L.hjo_denoise.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.hjo_tonemap.restype = C.c_float
        L.hjo_tonemap.argtypes = [C.c_float, C.c_int]
        L.hjo_create.restype = C.c_void_p
        L.hjo_create.argtypes = [C.c_void_p, C.c_int]
        L.hjo_destroy.argtypes = [C.c_void_p]
        L.hjo_render.restype = C.c_int
        L.hjo_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.hjo_sample.restype = C.c_int
        L.hjo_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
        L.hjo_trace_closest.restype = C.c_int
        L.hjo_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.hjo_trace_any.restype = C.c_int
        L.hjo_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """Owns numpy copies of a scene (dict of arrays, see scene_arrays()) and an hjo_ctx."""

    def __init__(self, arrays, math_mode=MATH_PORTABLE):
        a = {}
        a["vertices"] = np.ascontiguousarray(arrays["vertices"], dtype=np.float32).reshape(-1)
        a["normals"] = np.ascontiguousarray(arrays["normals"], dtype=np.float32).reshape(-1)
        a["texcoords"] = np.ascontiguousarray(arrays["texcoords"], dtype=np.float32).reshape(-1)
        a["indices"] = np.ascontiguousarray(arrays["indices"], dtype=np.uint32).reshape(-1)
        a["material_ids"] = np.ascontiguousarray(arrays["material_ids"], dtype=np.uint32).reshape(-1)
        a["prim_offsets"] = np.ascontiguousarray(arrays["prim_offsets"], dtype=np.uint32).reshape(-1)
        a["transforms"] = np.ascontiguousarray(arrays["transforms"], dtype=np.float32).reshape(-1)
        a["inv_transforms"] = np.ascontiguousarray(arrays["inv_transforms"], dtype=np.float32).reshape(-1)
        a["materials"] = np.ascontiguousarray(arrays["materials"], dtype=MATERIAL_DTYPE)
        a["light_prim_ids"] = np.ascontiguousarray(arrays["light_prim_ids"], dtype=np.uint32).reshape(-1)
        a["light_prim_emission"] = np.ascontiguousarray(arrays["light_prim_emission"], dtype=np.float32).reshape(-1)
        lut = arrays.get("lut_rgba")
        a["lut_rgba"] = None if lut is None else np.ascontiguousarray(lut, dtype=np.uint8)
        self.a = a
        s = Scene()
        s.n_tris = a["indices"].size // 3
        s.n_instances = a["prim_offsets"].size
        s.n_materials = a["materials"].size
        s.n_lights = a["light_prim_ids"].size
        for k in ("vertices", "normals", "texcoords", "indices", "material_ids", "prim_offsets", "transforms",
                  "inv_transforms", "materials", "light_prim_ids", "light_prim_emission"):
            setattr(s, k, a[k].ctypes.data if a[k].size else None)
        if a["lut_rgba"] is not None:
            s.lut_rgba = a["lut_rgba"].ctypes.data
            s.lut_h, s.lut_w = a["lut_rgba"].shape[:2]
        texs = arrays.get("textures") or []
        if texs:
            a["tex_px"] = [np.ascontiguousarray(t[0], dtype=np.uint8) for t in texs]
            self._texarr = (Texture * len(texs))()
            for i, (t, px) in enumerate(zip(texs, a["tex_px"])):
                self._texarr[i].rgba8 = px.ctypes.data
                self._texarr[i].height, self._texarr[i].width = px.shape[0], px.shape[1]
                self._texarr[i].srgb = int(t[1])
            s.textures = C.addressof(self._texarr)
            s.n_textures = len(texs)
        sky = arrays.get("sky_rgba")
        if sky is not None:
            a["sky_rgba"] = np.ascontiguousarray(sky, dtype=np.float32)
            s.sky_rgba = a["sky_rgba"].ctypes.data
            s.sky_h, s.sky_w = a["sky_rgba"].shape[:2]
        self.scene = s
        self.math_mode = math_mode
        self.ctx = lib().hjo_create(C.byref(s), math_mode)

    def close(self):
        if self.ctx:
            lib().hjo_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, params, nthreads=None, want_aovs=True):
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        n = params.width * params.height * 4
        color = np.zeros(n, dtype=np.float32)
        albedo = np.zeros(n, dtype=np.float32) if want_aovs else None
        normal = np.zeros(n, dtype=np.float32) if want_aovs else None
        st = Stats()
        rc = lib().hjo_render(self.ctx, C.byref(params), _ptr(color), _ptr(albedo), _ptr(normal), nthreads, C.byref(st))
        assert rc == 0
        shp = (params.height, params.width, 4)
        return (color.reshape(shp), None if albedo is None else albedo.reshape(shp),
                None if normal is None else normal.reshape(shp), st.as_dict())

    def sample(self, params, x, y, s):
        r, a, n = F3(), F3(), F3()
        lib().hjo_sample(self.ctx, C.byref(params), x, y, s, r, a, n)
        return np.array(r, dtype=np.float32), np.array(a, dtype=np.float32), np.array(n, dtype=np.float32)

    def sample_is_nan(self, params, x, y, s):
        """True when sample s of pixel (x, y) is NaN / Inf on the oracle (zeroed and counted by the guard, like the product's)."""
        r, a, n = F3(), F3(), F3()
        return lib().hjo_sample(self.ctx, C.byref(params), x, y, s, r, a, n) != 0

    def trace_closest(self, o, d, tmin=0.001, tmax=1e16, use_bvh=1):
        out = F3()
        p = lib().hjo_trace_closest(self.ctx, F3(*o), F3(*d), tmin, tmax, use_bvh, out)
        return p, np.array(out, dtype=np.float32)

    def trace_any(self, o, d, tmin, tmax, use_bvh=1):
        return lib().hjo_trace_any(self.ctx, F3(*o), F3(*d), tmin, tmax, use_bvh)


def make_params(width, height, spp, cam, frame=1, seed=1, integrator=INTEGRATOR_NEE, sky=(0.8, 0.8, 0.8),
                ibl_intensity=1.0, rect=None):
    p = Params()
    p.width, p.height, p.spp, p.frame, p.seed, p.integrator = width, height, spp, frame, seed, integrator
    p.cam_pos = F3(*cam["pos"])
    p.cam_dir = F3(*cam["dir"])
    p.cam_up = F3(*cam["up"])
    p.cam_right = F3(*cam["right"])
    p.cam_f = cam["f"]
    p.sky = F3(*sky)
    p.ibl_intensity = ibl_intensity
    if rect:
        p.x0, p.y0, p.x1, p.y1 = rect
    return p


def denoise(mode, color, albedo, normal):
    """hjo_denoise: the oracle's restatement of the denoise-mode replacement; float4 images (H, W, 4)."""
    color = np.ascontiguousarray(color, dtype=np.float32)
    albedo = np.ascontiguousarray(albedo, dtype=np.float32)
    normal = np.ascontiguousarray(normal, dtype=np.float32)
    h, w = color.shape[:2]
    ow, oh = (2 * w, 2 * h) if mode == 2 else (w, h)
    out = np.zeros((oh, ow, 4), dtype=np.float32)
    rc = lib().hjo_denoise(mode, w, h, color.ctypes.data, albedo.ctypes.data, normal.ctypes.data, out.ctypes.data, ow, oh)
    if rc != 0:
        raise RuntimeError("hjo_denoise failed: %d" % rc)
    return out
