"""Thin Python glue over libhenjou_hip.so (C-ABI: include/henjou_hip.h).

Python is plumbing here — tests, bench.py and the torch.distributed (RCCL) framebuffer exchange.  The product is
the C++/HIP library; this module adds no compute path and no fallback: if the library is missing or no MI355X is
present, the calls raise.

`Renderer` mirrors the reference's `class Renderer` (renderer/renderer.h:900-1318): loadRenderOption,
loadGLTFfile, build, and a per-frame render that returns the linear float4 AOVs.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("HJR_LIB") or os.path.join(PKG_DIR, "libhenjou_hip.so")  # HJR_LIB: kernel-variant experiments only
ASSETS = os.path.join(PKG_DIR, "assets")

INTEGRATOR_NEE, INTEGRATOR_PT, INTEGRATOR_MIS = 0, 1, 2
MODE_DEFAULT, MODE_DENOISE, MODE_DENOISE_UPSCALE2X, MODE_DEBUG = 0, 1, 2, 3  # render_option.h:38-43
FLAG_STATS, FLAG_ZERO_UNOWNED, FLAG_PACKED, FLAG_FAST_MATH = 1, 2, 4, 8


class HjrError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("basecolor", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("sheen", C.c_float), ("clearcoat", C.c_float), ("ior", C.c_float),
                ("transmission", C.c_float), ("emission", C.c_float * 3), ("is_light", C.c_int32),
                ("ideal_specular", C.c_int32), ("is_thinfilm", C.c_int32), ("basecolor_tex", C.c_int32),
                ("metallic_roughness_tex", C.c_int32), ("normal_tex", C.c_int32), ("emission_tex", C.c_int32),
                ("_reserved", C.c_int32)]


MATERIAL_DTYPE = np.dtype([("basecolor", "<f4", 3), ("metallic", "<f4"), ("roughness", "<f4"), ("sheen", "<f4"),
                           ("clearcoat", "<f4"), ("ior", "<f4"), ("transmission", "<f4"), ("emission", "<f4", 3),
                           ("is_light", "<i4"), ("ideal_specular", "<i4"), ("is_thinfilm", "<i4"),
                           ("basecolor_tex", "<i4"), ("metallic_roughness_tex", "<i4"), ("normal_tex", "<i4"),
                           ("emission_tex", "<i4"), ("_reserved", "<i4")])


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("srgb", C.c_int32),
                ("_reserved", C.c_int32)]


class _Sized(C.Structure):
    """Sized struct of the C-ABI (include/henjou_hip.h): struct_size = sizeof(this struct), set on construction (HJR_INIT)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)


class SceneView(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("n_instances", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_lights", C.c_uint32), ("n_animations", C.c_uint32),
                ("n_textures", C.c_uint32),
                ("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p),
                ("indices", C.c_void_p), ("material_ids", C.c_void_p), ("prim_offset", C.c_void_p),
                ("geometry_index_offset", C.c_void_p), ("geometry_index_count", C.c_void_p),
                ("instance_animation_id", C.c_void_p), ("materials", C.c_void_p),
                ("light_prim_ids", C.c_void_p), ("light_prim_emission", C.c_void_p), ("textures", C.c_void_p)]


class RenderOption(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("image_name", C.c_char * 256),
                ("image_directory", C.c_char * 512), ("max_spp", C.c_uint32), ("gltf_path", C.c_char * 512),
                ("gltf_name", C.c_char * 256), ("fps", C.c_uint32), ("start_frame", C.c_uint32),
                ("end_frame", C.c_uint32), ("time_limit", C.c_float), ("allow_camera_animation", C.c_int32),
                ("camera_fov", C.c_float), ("camera_position", C.c_float * 3), ("camera_direction", C.c_float * 3),
                ("camera_animation_id", C.c_int32), ("render_mode", C.c_int32), ("ptxfile_path", C.c_char * 512),
                ("use_IBL", C.c_int32), ("IBL_path", C.c_char * 512), ("IBL_intensity", C.c_float),
                ("scene_sky_default", C.c_float * 3), ("use_date", C.c_int32), ("save_renderOption", C.c_int32),
                ("LUT_path", C.c_char * 512), ("seed", C.c_uint32), ("integrator", C.c_int32),
                ("devices", C.c_uint32), ("tile", C.c_uint32), ("serial_io", C.c_int32), ("fast_math", C.c_int32), ("force_rebuild", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("f", C.c_float)]

    def as_dict(self):
        return {"pos": list(self.pos), "dir": list(self.dir), "up": list(self.up), "right": list(self.right),
                "f": float(self.f)}


class Params(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("frame", C.c_uint32),
                ("seed", C.c_uint32), ("integrator", C.c_uint32), ("camera", Camera), ("sky", C.c_float * 3),
                ("ibl_intensity", C.c_float), ("rank", C.c_uint32), ("world_size", C.c_uint32),
                ("flags", C.c_uint32)]


class Stats(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("_pad0", C.c_uint32)] + [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "box_tests_closest",
                                           "tri_tests_closest", "box_tests_shadow", "tri_tests_shadow",
                                           "shaded_hits", "light_samples", "nan_samples")] + \
               [("last_kernel_ms", C.c_float), ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("n_triangles", C.c_uint32), ("lds_mode", C.c_uint32), ("stack_need", C.c_uint32),
                ("stack_lds_entries", C.c_uint32), ("pipeline", C.c_uint32), ("stack_overflow_pushes", C.c_uint64),
                ("nan_located", C.c_uint32), ("fast_math", C.c_uint32), ("nan_where", (C.c_uint32 * 3) * 8)]

    def as_dict(self):
        d = {n: (float(getattr(self, n)) if n == "last_kernel_ms" else int(getattr(self, n)))
             for n, _ in self._fields_ if n not in ("struct_size", "_pad0", "nan_where")}
        d["nan_where"] = [tuple(int(v) for v in self.nan_where[i]) for i in range(int(self.nan_located))]  # (x, y, sample)
        return d


_lib = None


def lib():
    """Loads libhenjou_hip.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HjrError("libhenjou_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C henjou-renderer_amd`; there is no fallback path")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname libamdhip64.so.7) and asks for
        # it by file name, so if /opt/rocm's copy were loaded first torch would load a second runtime and then see no GPU.
        # Importing torch first makes libhenjou_hip.so's NEEDED libamdhip64.so.7 resolve to the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.hjr_last_error.restype = C.c_char_p
        for name, args in {
            "hjr_load_render_option": [C.c_char_p, C.c_void_p],
            "hjr_scene_load_gltf": [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p],
            "hjr_scene_get_view": [C.c_void_p, C.c_void_p],
            "hjr_scene_eval_transforms": [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p],
            "hjr_scene_eval_camera": [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p],
            "hjr_load_png_rgba8": [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "hjr_load_hdr_rgba32f": [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "hjr_set_sky": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
            "hjr_create": [C.c_int, C.c_void_p],
            "hjr_upload_scene": [C.c_void_p, C.c_void_p],
            "hjr_set_transforms": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32],
            "hjr_set_lut": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
            "hjr_render": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "hjr_render_device": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "hjr_synchronize": [C.c_void_p],
            "hjr_get_stats": [C.c_void_p, C.c_void_p],
            "hjr_preview_device": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p],
            "hjr_set_option": [C.c_void_p, C.c_char_p, C.c_int],
            "hjr_get_option": [C.c_void_p, C.c_char_p, C.c_void_p],
            "hjr_float4_to_srgb8": [C.c_void_p, C.c_void_p, C.c_uint32],
            "hjr_tonemap_to_srgb8": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int],
            "hjr_write_png": [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int],
            "hjr_write_pfm": [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32],
            "hjr_render_file": [C.c_char_p, C.c_int],
            "hjr_load_image_rgba8": [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "hjr_denoise": [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32],
            "hjr_denoise_device": [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p],
            "hjr_render_denoised": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32],
            "hjr_pack_tiles": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p],
            "hjr_unpack_tiles": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p],
            "hjr_pack_tiles_device": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p],
            "hjr_unpack_tiles_device": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p],
            "hjr_selftest_stack16": [],
        }.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = args
        L.hjr_owned_tiles.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.hjr_owned_tiles.restype = C.c_uint32
        L.hjr_scene_free.argtypes = [C.c_void_p]
        L.hjr_scene_free.restype = None
        L.hjr_destroy.argtypes = [C.c_void_p]
        L.hjr_destroy.restype = None
        L.hjr_free.argtypes = [C.c_void_p]
        L.hjr_free.restype = None
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HjrError("%s failed (%d): %s" % (what, rc, lib().hjr_last_error().decode("utf-8", "replace")))


def _np(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def load_render_option(path):
    """load_json (loader/render_json_loader.h:78-228)."""
    opt = RenderOption()
    _check(lib().hjr_load_render_option(os.fsencode(path), C.byref(opt)), "hjr_load_render_option")
    return opt


def load_png(path):
    p = C.c_void_p()
    w, h = C.c_int(), C.c_int()
    _check(lib().hjr_load_png_rgba8(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)), "hjr_load_png_rgba8")
    a = _np(p.value, w.value * h.value * 4, np.uint8).reshape(h.value, w.value, 4)
    lib().hjr_free(p)
    return a


def load_image(path):
    """Material texture decode: PNG or baseline JPEG by signature (hjr_load_image_rgba8) -> uint8 [h, w, 4]."""
    p = C.c_void_p()
    w, h = C.c_int(), C.c_int()
    _check(lib().hjr_load_image_rgba8(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)), "hjr_load_image_rgba8")
    a = _np(p.value, w.value * h.value * 4, np.uint8).reshape(h.value, w.value, 4)
    lib().hjr_free(p)
    return a


def load_hdr(path):
    """Radiance .hdr -> float32 [h, w, 4] (a = 0), as HDRTexture holds it (renderer/texture.h:67-88)."""
    p = C.c_void_p()
    w, h = C.c_int(), C.c_int()
    _check(lib().hjr_load_hdr_rgba32f(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)), "hjr_load_hdr_rgba32f")
    a = _np(p.value, w.value * h.value * 4, np.float32).reshape(h.value, w.value, 4)
    lib().hjr_free(p)
    return a


def write_png(path, rgba8, flip_y=True):
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    _check(lib().hjr_write_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], 1 if flip_y else 0), "hjr_write_png")


def float4_to_srgb8(rgba):
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.zeros(a.shape, dtype=np.uint8)
    _check(lib().hjr_float4_to_srgb8(a.ctypes.data, out.ctypes.data, a.size // 4), "hjr_float4_to_srgb8")
    return out


TONEMAP_NONE, TONEMAP_UCHIMURA, TONEMAP_ACES = 0, 1, 2


def tonemap_to_srgb8(rgba, tonemap):
    """kernel/color.h tonemapper, then toSRGB + quantise (renderer.h:73-101)."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.zeros(a.shape, dtype=np.uint8)
    _check(lib().hjr_tonemap_to_srgb8(a.ctypes.data, out.ctypes.data, a.size // 4, tonemap), "hjr_tonemap_to_srgb8")
    return out


class Scene:
    """Owning handle of a loaded glTF scene (SceneData, renderer/scene.h:19-36)."""

    def __init__(self, directory, filename, opt):
        self._h = C.c_void_p()
        _check(lib().hjr_scene_load_gltf(os.fsencode(directory), os.fsencode(filename), C.byref(opt), C.byref(self._h)),
               "hjr_scene_load_gltf")
        self.view = SceneView()
        _check(lib().hjr_scene_get_view(self._h, C.byref(self.view)), "hjr_scene_get_view")

    def close(self):
        if self._h:
            lib().hjr_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transforms(self, time):
        n = self.view.n_instances
        m = np.zeros((n, 12), dtype=np.float32)
        inv = np.zeros((n, 12), dtype=np.float32)
        _check(lib().hjr_scene_eval_transforms(self._h, C.c_float(time), m.ctypes.data, inv.ctypes.data),
               "hjr_scene_eval_transforms")
        return m, inv

    def camera(self, opt, time):
        cam = Camera()
        _check(lib().hjr_scene_eval_camera(self._h, C.byref(opt), C.c_float(time), C.byref(cam)), "hjr_scene_eval_camera")
        return cam

    def arrays(self, time=None):
        """numpy copies of the SceneData columns (+ transforms at `time`) — the oracle's input in tests/bench."""
        v = self.view
        a = {
            "vertices": _np(v.vertices, v.n_vertices * 3, np.float32),
            "normals": _np(v.normals, v.n_vertices * 3, np.float32),
            "texcoords": _np(v.texcoords, v.n_vertices * 2, np.float32),
            "indices": _np(v.indices, v.n_triangles * 3, np.uint32),
            "material_ids": _np(v.material_ids, v.n_triangles, np.uint32),
            "prim_offsets": _np(v.prim_offset, v.n_instances, np.uint32),
            "geometry_index_offset": _np(v.geometry_index_offset, v.n_instances, np.uint32),
            "geometry_index_count": _np(v.geometry_index_count, v.n_instances, np.uint32),
            "instance_animation_id": _np(v.instance_animation_id, v.n_instances, np.uint32),
            "materials": _np(v.materials, v.n_materials, MATERIAL_DTYPE),
            "light_prim_ids": _np(v.light_prim_ids, v.n_lights, np.uint32),
            "light_prim_emission": _np(v.light_prim_emission, v.n_lights * 3, np.float32),
        }
        texs = []
        if v.n_textures:
            arr = (Texture * v.n_textures).from_address(v.textures)
            for t in arr:
                texs.append((_np(t.rgba8, t.width * t.height * 4, np.uint8).reshape(t.height, t.width, 4), int(t.srgb)))
        a["textures"] = texs
        if time is not None:
            a["transforms"], a["inv_transforms"] = self.transforms(time)
        return a


class Device:
    """One hjr_ctx (one per GPU / per process)."""

    def __init__(self, ordinal=0):
        self._h = C.c_void_p()
        _check(lib().hjr_create(ordinal, C.byref(self._h)), "hjr_create")

    def close(self):
        if self._h:
            lib().hjr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, view):
        _check(lib().hjr_upload_scene(self._h, C.byref(view)), "hjr_upload_scene")

    def upload_arrays(self, a):
        """Upload from numpy arrays (keys as Scene.arrays())."""
        self._keep = {k: np.ascontiguousarray(a[k]) for k in ("vertices", "normals", "texcoords", "indices", "material_ids",
                                                              "prim_offsets", "materials", "light_prim_ids", "light_prim_emission")}
        k = self._keep
        v = SceneView()
        v.n_vertices = k["vertices"].size // 3
        v.n_triangles = k["indices"].size // 3
        v.n_instances = k["prim_offsets"].size
        v.n_materials = k["materials"].size
        v.n_lights = k["light_prim_ids"].size
        texs = a.get("textures") or []
        if texs:
            self._keep_tex = [np.ascontiguousarray(t[0], dtype=np.uint8) for t in texs]
            self._keep_texarr = (Texture * len(texs))()
            for i, (t, px) in enumerate(zip(texs, self._keep_tex)):
                self._keep_texarr[i].rgba8 = px.ctypes.data
                self._keep_texarr[i].height, self._keep_texarr[i].width = px.shape[0], px.shape[1]
                self._keep_texarr[i].srgb = int(t[1])
            v.n_textures = len(texs)
            v.textures = C.addressof(self._keep_texarr)
        for name, key in (("vertices", "vertices"), ("normals", "normals"), ("texcoords", "texcoords"), ("indices", "indices"),
                          ("material_ids", "material_ids"), ("prim_offset", "prim_offsets"), ("materials", "materials"),
                          ("light_prim_ids", "light_prim_ids"), ("light_prim_emission", "light_prim_emission")):
            setattr(v, name, k[key].ctypes.data if k[key].size else None)
        self.upload_scene(v)

    def set_transforms(self, m, inv):
        m = np.ascontiguousarray(m, dtype=np.float32)
        inv = np.ascontiguousarray(inv, dtype=np.float32)
        _check(lib().hjr_set_transforms(self._h, m.ctypes.data, inv.ctypes.data, m.size // 12), "hjr_set_transforms")

    def set_lut(self, rgba):
        if rgba is None:
            _check(lib().hjr_set_lut(self._h, None, 0, 0), "hjr_set_lut")
            return
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        _check(lib().hjr_set_lut(self._h, a.ctypes.data, a.shape[1], a.shape[0]), "hjr_set_lut")

    def set_sky(self, rgba32f):
        """Equirect IBL (float RGBA [h, w, 4]); None restores the constant scene_sky_default sky."""
        if rgba32f is None:
            _check(lib().hjr_set_sky(self._h, None, 0, 0), "hjr_set_sky")
            return
        a = np.ascontiguousarray(rgba32f, dtype=np.float32)
        _check(lib().hjr_set_sky(self._h, a.ctypes.data, a.shape[1], a.shape[0]), "hjr_set_sky")

    def render(self, params, want_aovs=True):
        """Synchronous render into host arrays (hjr_render)."""
        shp = (params.height, params.width, 4)
        color = np.zeros(shp, dtype=np.float32)
        albedo = np.zeros(shp, dtype=np.float32) if want_aovs else None
        normal = np.zeros(shp, dtype=np.float32) if want_aovs else None
        _check(lib().hjr_render(self._h, C.byref(params), color.ctypes.data,
                                albedo.ctypes.data if want_aovs else None, normal.ctypes.data if want_aovs else None), "hjr_render")
        return color, albedo, normal

    def render_device(self, params, d_color, d_albedo=None, d_normal=None, stream=None):
        """Asynchronous render into device pointers (ints, e.g. torch tensor .data_ptr()) on a hipStream_t (int)."""
        _check(lib().hjr_render_device(self._h, C.byref(params), C.c_void_p(d_color),
                                       C.c_void_p(d_albedo) if d_albedo else None, C.c_void_p(d_normal) if d_normal else None,
                                       C.c_void_p(stream) if stream else None), "hjr_render_device")

    def denoise(self, mode, color, albedo=None, normal=None):
        """OptixDenoiserManager::denoise() replacement on host float4 images (hjr_denoise); returns AOV_Output."""
        color = np.ascontiguousarray(color, dtype=np.float32)
        h, w = color.shape[:2]
        ow, oh = (2 * w, 2 * h) if mode == MODE_DENOISE_UPSCALE2X else (w, h)
        a = None if albedo is None else np.ascontiguousarray(albedo, dtype=np.float32)
        n = None if normal is None else np.ascontiguousarray(normal, dtype=np.float32)
        out = np.zeros((oh, ow, 4), dtype=np.float32)
        _check(lib().hjr_denoise(self._h, mode, w, h, color.ctypes.data, None if a is None else a.ctypes.data,
                                 None if n is None else n.ctypes.data, out.ctypes.data, ow, oh), "hjr_denoise")
        return out

    def render_denoised(self, params, mode, out_w=None, out_h=None):
        """One frame in a render mode, AOVs kept on the device (hjr_render_denoised); returns AOV_Output."""
        ow = out_w if out_w is not None else (2 * params.width if mode == MODE_DENOISE_UPSCALE2X else params.width)
        oh = out_h if out_h is not None else (2 * params.height if mode == MODE_DENOISE_UPSCALE2X else params.height)
        out = np.zeros((oh, ow, 4), dtype=np.float32)
        _check(lib().hjr_render_denoised(self._h, C.byref(params), mode, out.ctypes.data, ow, oh), "hjr_render_denoised")
        return out

    def synchronize(self):
        _check(lib().hjr_synchronize(self._h), "hjr_synchronize")

    def preview_device(self, d_color, width, height, tonemap, d_rgba8, stream=None):
        """hjr_preview_device: the raygen's 8-bit preview image from a float4 colour image, device pointers (ints)."""
        _check(lib().hjr_preview_device(self._h, C.c_void_p(d_color), width, height, tonemap, C.c_void_p(d_rgba8),
                                        C.c_void_p(stream) if stream else None), "hjr_preview_device")

    def set_option(self, key, value):
        """hjr_set_option: tuning / test option of this context (-1 restores the default); layout options act at the next set_transforms."""
        _check(lib().hjr_set_option(self._h, key.encode(), int(value)), "hjr_set_option")

    def get_option(self, key):
        v = C.c_int()
        _check(lib().hjr_get_option(self._h, key.encode(), C.byref(v)), "hjr_get_option")
        return v.value

    def stats(self):
        st = Stats()
        _check(lib().hjr_get_stats(self._h, C.byref(st)), "hjr_get_stats")
        return st.as_dict()


def make_params(width, height, spp, camera, frame=1, seed=1, integrator=INTEGRATOR_NEE, sky=(0.8, 0.8, 0.8),
                ibl_intensity=1.0, rank=0, world_size=1, flags=0):
    p = Params()
    p.width, p.height, p.spp, p.frame, p.seed, p.integrator = width, height, spp, frame, seed, integrator
    if isinstance(camera, Camera):
        p.camera = camera
    else:
        p.camera.pos = (C.c_float * 3)(*camera["pos"])
        p.camera.dir = (C.c_float * 3)(*camera["dir"])
        p.camera.up = (C.c_float * 3)(*camera["up"])
        p.camera.right = (C.c_float * 3)(*camera["right"])
        p.camera.f = camera["f"]
    p.sky = (C.c_float * 3)(*sky)
    p.ibl_intensity = ibl_intensity
    p.rank, p.world_size, p.flags = rank, world_size, flags
    return p


def owned_tile_mask(width, height, rank, world_size, tile=8):
    """Boolean [height, width] mask of the pixels rank `rank` renders: 8x8 tiles, tile (tx, ty) has id ty * tiles_x + (tx + ty) % tiles_x
    (rows rotated so that a rank's tiles run along diagonals, csrc/hjr_layout.h) and belongs to rank id % world_size."""
    tx = (np.arange(width) // tile)[None, :]
    ty = (np.arange(height) // tile)[:, None]
    tiles_x = (width + tile - 1) // tile
    return ((ty * tiles_x + (tx + ty) % tiles_x) % world_size) == rank


def exchange_framebuffer(fb, dst=0):
    """Full-frame form of the multi-GPU exchange (what north_star names): every rank holds its own 8x8 tiles and zeros
    elsewhere (HJR_FLAG_ZERO_UNOWNED); a SUM reduce onto `dst` assembles the frame.  Adding zeros is exact in IEEE arithmetic,
    so the result is bit-identical to the 1-GPU image.  `fb` is a torch tensor (CUDA -> RCCL over xGMI; CPU -> gloo in the
    tests).  gather_tiles below moves 1 / world of these bytes and is what bench.py and henjou_cli use."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def owned_tiles(width, height, rank, world_size):
    return int(lib().hjr_owned_tiles(width, height, rank, world_size))


def pack_tiles(frame, rank, world_size):
    """Host form of the packed layout: [owned tile][64] float4 of `rank` from a row-major float4 frame (hjr_pack_tiles)."""
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    h, w = frame.shape[:2]
    out = np.zeros((owned_tiles(w, h, rank, world_size), 64, 4), dtype=np.float32)
    _check(lib().hjr_pack_tiles(frame.ctypes.data, w, h, rank, world_size, out.ctypes.data), "hjr_pack_tiles")
    return out


def unpack_tiles(packed, frame, rank, world_size):
    """Scatters one rank's packed tiles into `frame` (row-major float4, modified in place; hjr_unpack_tiles)."""
    packed = np.ascontiguousarray(packed, dtype=np.float32)
    h, w = frame.shape[:2]
    assert frame.dtype == np.float32 and frame.flags["C_CONTIGUOUS"]
    _check(lib().hjr_unpack_tiles(packed.ctypes.data, w, h, rank, world_size, frame.ctypes.data), "hjr_unpack_tiles")
    return frame


def gather_tiles(packed, width, height, device=None, dst=0, frame=None):
    """The multi-GPU exchange sized by ownership: every rank contributes its packed tiles
    ([max owned tiles][64][4] float32 torch tensor, the same shape on every rank: ranks owning one tile fewer pad), rank `dst`
    receives the world_size blocks (RCCL gather = point-to-point sends over xGMI, all peers in parallel) and scatters each into
    the row-major frame: on a CUDA tensor with hjr_unpack_tiles_device (`device` = the rank's hjr Device), on a CPU tensor
    (gloo, tests) with hjr_unpack_tiles.  Returns the assembled frame on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    blocks = None
    if world > 1:
        if rank == dst:
            blocks = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=blocks, dst=dst)
    else:
        blocks = [packed]
    if rank != dst:
        return None
    if frame is None:
        frame = torch.zeros((height, width, 4), dtype=torch.float32, device=packed.device)
    for r, blk in enumerate(blocks):
        if blk.is_cuda:
            _check(lib().hjr_unpack_tiles_device(device._h, C.c_void_p(blk.data_ptr()), width, height, r, world, C.c_void_p(frame.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hjr_unpack_tiles_device")
        else:
            unpack_tiles(blk.numpy(), frame.numpy(), r, world)
    return frame


class Renderer:
    """Mirror of the reference's `class Renderer` (renderer/renderer.h:900-1318) on top of the C-ABI."""

    def __init__(self, device_ordinal=0):
        self.device_ordinal = device_ordinal
        self.render_option = None
        self.scene = None
        self.device = None

    def loadRenderOption(self, path):  # renderer.h:1041-1051
        self.render_option = load_render_option(path)
        return True

    def setRenderOption(self, opt):  # renderer.h:992-995
        self.render_option = opt

    def loadGLTFfile(self, filepath, filename):  # renderer.h:1005-1013
        self.scene = Scene(filepath, filename, self.render_option)
        return True

    def build(self):  # renderer.h:1015-1039
        self.device = Device(self.device_ordinal)
        self.device.upload_scene(self.scene.view)
        lut_path = self.render_option.LUT_path.decode()
        self.lut = None
        if lut_path and os.path.exists(lut_path):
            self.lut = load_png(lut_path)
            self.device.set_lut(self.lut)
        ibl = self.render_option.IBL_path.decode()
        if self.render_option.use_IBL and ibl and os.path.exists(ibl):  # setSky (renderer.h:802-851)
            self.device.set_sky(load_hdr(ibl))

    def frame_params(self, frame, spp=None, rank=0, world_size=1, flags=0):
        o = self.render_option
        time = frame / float(o.fps)
        cam = self.scene.camera(o, time)
        return make_params(o.image_width, o.image_height, spp or o.max_spp, cam, frame=frame, seed=o.seed,
                           integrator=o.integrator, sky=tuple(o.scene_sky_default), ibl_intensity=o.IBL_intensity,
                           rank=rank, world_size=world_size, flags=flags), time

    def render_frame(self, frame, spp=None):
        """One iteration of the frame loop (renderer.h:1126-1245): IAS update, camera, launch; returns float4 AOVs."""
        p, time = self.frame_params(frame, spp)
        m, inv = self.scene.transforms(time)
        self.device.set_transforms(m, inv)
        return self.device.render(p)

    def initializeAndRender(self, render_option_path):  # renderer.h:1053-1317
        _check(lib().hjr_render_file(os.fsencode(render_option_path), self.device_ordinal), "hjr_render_file")
        return True
