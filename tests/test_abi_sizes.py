"""Sized structs of the C-ABI under AddressSanitizer / UBSan on the CPU: callers with shorter (older) and longer (newer) structs than the
library's, each allocated with exactly struct_size bytes (tests/native/abi_driver.cpp).  The round-2 library overran a caller's
`hjr_stats` that predated its appended fields (`*** stack smashing detected ***` in a tools/kbench run); this is the test of the rule
that replaced "append fields and hope" (include/henjou_hip.h)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from scene_util import ROOT, hjr


def test_short_and_long_structs_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "abi_driver")
    host = os.path.join(ROOT, "henjou-renderer_amd", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "native", "abi_driver.cpp"), os.path.join(host, "capi.cpp"), os.path.join(host, "loaders.cpp"),
                           os.path.join(host, "image_io.cpp"), os.path.join(host, "jpeg.cpp"), "-o", exe, "-lz", "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, hjr.ASSETS, "render_option_c1.json"], capture_output=True, text=True, env=env, timeout=300, cwd=hjr.ASSETS)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    assert "abi_driver ok" in p.stdout


def test_shipped_library_honours_struct_size():
    """The same rule through the shipped libhenjou_hip.so (ctypes): a short hjr_render_option inside a guarded buffer."""
    L = hjr.lib()
    short = hjr.RenderOption.seed.offset  # the round-1 struct ended before the Henjou_HIP section
    buf = (C.c_ubyte * (short + 256))()
    C.memset(buf, 0xEE, short + 256)
    C.memmove(buf, C.byref(C.c_uint32(short)), 4)
    cwd = os.getcwd()
    os.chdir(hjr.ASSETS)
    try:
        assert L.hjr_load_render_option(b"render_option_c1.json", C.byref(buf)) == 0
    finally:
        os.chdir(cwd)
    got = hjr.RenderOption.from_buffer_copy(bytes(buf)[:C.sizeof(hjr.RenderOption)] if short + 256 >= C.sizeof(hjr.RenderOption) else bytes(buf) + bytes(4096))
    assert got.struct_size == short and got.image_width == 256 and got.max_spp == 16
    assert all(b == 0xEE for b in bytes(buf)[short:]), "bytes behind the caller's struct_size were written"
    zero = (C.c_ubyte * C.sizeof(hjr.RenderOption))()
    assert L.hjr_load_render_option(b"render_option_c1.json", C.byref(zero)) == -1
    assert b"struct_size" in L.hjr_last_error()


@pytest.mark.gpu
def test_device_entry_points_honour_struct_size():
    """hjr_get_stats with the round-1 hjr_stats (the struct a stale tools/kbench had on its stack) and hjr_render with an hjr_params
    from before the tile shard, both inside guarded buffers, on the GPU."""
    from scene_util import Cornell
    sc = Cornell()
    dev = sc.device()
    p = sc.hjr_params(64, 64, 4)
    color, _, _ = dev.render(p)
    L = hjr.lib()
    s1 = hjr.Stats.lds_mode.offset
    buf = (C.c_ubyte * (s1 + 128))()
    C.memset(buf, 0xEE, s1 + 128)
    C.memmove(buf, C.byref(C.c_uint32(s1)), 4)
    assert L.hjr_get_stats(dev._h, C.byref(buf)) == 0
    assert all(b == 0xEE for b in bytes(buf)[s1:]), "hjr_get_stats wrote behind the caller's struct_size"
    st = hjr.Stats.from_buffer_copy(bytes(buf)[:s1] + bytes(C.sizeof(hjr.Stats) - s1))
    assert st.n_triangles == 984 and st.last_kernel_ms > 0
    # a short hjr_params (no rank / world_size / flags): same picture as the full struct with rank 0 of 1
    p1 = hjr.Params.rank.offset
    pb = (C.c_ubyte * p1)()
    C.memmove(pb, C.byref(p), p1)
    C.memmove(pb, C.byref(C.c_uint32(p1)), 4)
    out = np.zeros((64, 64, 4), np.float32)
    assert L.hjr_render(dev._h, C.byref(pb), out.ctypes.data_as(C.c_void_p), None, None) == 0
    assert np.array_equal(out, color)
