#!/usr/bin/env python3
"""bench.py — Msamples/s of the Henjou hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one frame: the persistent HIP megakernel renders this rank's 8x8 pixel tiles of
the BASELINE configs[1] workload (bundled cornelbox.gltf, 1920x1080, 256 spp, NEE integrator, synthetic = the bundled
scene, no external data), and for N > 1 the ranks' packed colour tiles (1 / N of the frame each) are gathered onto rank 0 with one RCCL
collective over xGMI and scattered into the frame there.
Scene, BVH and all buffers are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080 --spp 256]
    N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
            or plainly `python bench.py --gpus N ...`: without WORLD_SIZE in the environment bench.py starts that launcher itself as a
            child process (before anything touches the GPU), relays its output and exits with its code.

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` is the rate of the launch the reference's raygen
performs (colour + albedo + normal AOVs, renderer/renderer.h:1222-1224); `color_only` carries the rate of the lean colour-only
instantiation next to it.  Plus
  "roofline":     the bound is chosen per launch from hardware counters collected IN THIS RUN (rocprofv3 --pmc passes over
                  tools/kbench, the same library and workload, started before this process touches the GPU):
                    * "valu": VALU lane-slot utilisation in the peak's units: every VALU wave instruction x its active lanes counted as
                      one 2-FLOP operation per second, against the 157.3 TFLOP/s vector peak of MI355X_MICROARCH.md (= 64 lanes x every
                      issue slot x 2).  It is an UPPER bound on the useful-FLOP fraction: moves, compares, integer and address
                      arithmetic count like an fma.  `roofline.flops` next to it counts only the fp32 arithmetic instructions
                      (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32; an fma = 2 FLOP) against the same peak;
                    * "hbm": the COUNTER bytes (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included) against 8 TB/s;
                  whichever fraction is larger is the roof the launch is closer to and becomes `bound` / `frac` (both <= 1 by
                  construction, both reported under `counters`).
                  SURVEY.md section 8d's algorithmic-bytes figure stays as a side field (it counts reads that LDS serves).
  "cpu_baseline": the CPU oracle (a from-scratch port; the reference has no CPU path) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sample(st, spp):
    """SURVEY.md §8(d) / BASELINE.md §4: bytes the algorithm must touch per sample, from the kernel's own counters.
    32 B per box test (AABB + link), 36 B per triangle test, 232 B per shaded hit, 192 B per light sample,
    52 B of AOV write-out per pixel."""
    n = max(st["samples"], 1)
    b = (st["box_tests_closest"] * 32 + st["tri_tests_closest"] * 36 + st["box_tests_shadow"] * 32 +
         st["tri_tests_shadow"] * 36 + st["shaded_hits"] * 232 + st["light_samples"] * 192) / n
    return b + 52.0 / spp


VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector) = 256 CUs x 4 SIMDs x 32 lanes x 2 FLOP x 2.4 GHz
PMC_PASSES = [  # separate passes: SQ has 8 slots, FETCH_SIZE / WRITE_SIZE do not fit one TCC pass (MI355X_MICROARCH.md, PMC slots)
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]
PMC_FLOP_PASS = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT",
                 "SQ_INSTS_SALU", "SQ_INSTS_LDS"]


def collect_pmc(kbench_args, env=None, timeout=150, passes=None):
    """Runs tools/kbench (one warm-up + one measured frame of the bench workload through the same libhenjou_hip.so) under
    `rocprofv3 --pmc`, one pass per counter group, and returns {counter: value per FRAME summed over the product's kernels}.
    The program itself follows `--` (no shell / env / launcher hop).  Raises on any failure: the caller decides what to report."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    kb = os.path.join(ROOT, "tools", "kbench")
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(kb):
        raise RuntimeError("tools/kbench is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    out = {}
    frames = 2  # kbench --reps 1 renders the frame twice (warm-up + 1)
    e = dict(os.environ)
    e.update(env or {})
    e["TMPDIR"] = "/tmp"
    for counters in (passes or PMC_PASSES):
        d = tempfile.mkdtemp(prefix="hjr_pmc_", dir="/tmp")
        try:
            cmd = [rp, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", kb] + kbench_args
            r = subprocess.run(cmd, cwd=os.path.join(ROOT, "henjou-renderer_amd", "assets"), env=e, capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 pass %s failed (rc %d): %s" % (counters, r.returncode, (r.stderr or r.stdout)[-400:]))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError("rocprofv3 pass %s wrote no counter_collection.csv" % counters)
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if "hjr_" in row["Kernel_Name"]:
                        acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for c in counters:
                if c not in acc:
                    raise RuntimeError("counter %s missing from the rocprofv3 output" % c)
                out[c] = acc[c] / frames
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def stress_secondary(lib_path, spheres, segments, env=None):
    """Secondary block of the default bench line: the generated ~1 M-triangle scene (BVH4 read from L2 / Infinity Cache / HBM instead of
    LDS) at 1920x1080 x 64 spp NEE through the same library, in tools/kbench child processes: kernel ms from the library's HIP events,
    fabric bytes from separate FETCH_SIZE / WRITE_SIZE counter passes of the same command."""
    import re
    import subprocess
    import tempfile
    sdir = os.path.join(tempfile.gettempdir(), "hjr_stress_%d_%d_sec" % (spheres, segments))
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_stress_scene.py"), sdir, "--spheres", str(spheres), "--segments", str(segments)],
                         capture_output=True, text=True, check=True)
    tri = re.search(r"= (\d+) triangles", gen.stdout)
    cfg = os.path.join(sdir, "render_option_stress.json")
    e = dict(os.environ)
    e.update(env or {})
    kb = os.path.join(ROOT, "tools", "kbench")
    r = subprocess.run([kb, lib_path, cfg, "--reps", "3"], cwd=os.path.join(ROOT, "henjou-renderer_amd", "assets"), env=e, capture_output=True, text=True, timeout=300)
    m = re.search(r" (\d+)x(\d+)x(\d+) integ .*kernel ms min ([0-9.]+) mean ([0-9.]+)", r.stdout)
    if r.returncode != 0 or not m:
        raise RuntimeError("kbench on the stress scene failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
    w, h, spp, ms = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(5))
    c = collect_pmc([lib_path, cfg, "--reps", "1"], env=env, timeout=300, passes=[["FETCH_SIZE"], ["WRITE_SIZE"]])
    traffic = c["FETCH_SIZE"] * 1024.0 * 2.0 + c["WRITE_SIZE"] * 1024.0
    gbps = traffic / (ms * 1e-3) / 1e9
    # what the algorithm has to touch: the counting variant's own box / triangle / hit / light-sample counts at 1/8 of the samples, priced with
    # the byte sizes of THIS library's records (hjr_layout.h: 64-byte compressed BVH4 node per 4 box tests, 48-byte triangle, 64 + 80-byte
    # shading + material record per shaded hit, 96-byte light record per light sample, 16 bytes of output per pixel)
    algorithmic = None
    rs = subprocess.run([kb, lib_path, cfg, "--reps", "1", "--stats", "--spp", str(max(spp // 8, 1))], cwd=os.path.join(ROOT, "henjou-renderer_amd", "assets"), env=e,
                        capture_output=True, text=True, timeout=300)
    ms_ = re.search(r"samples (\d+) closest (\d+) shadow (\d+) box_c (\d+) tri_c (\d+) box_s (\d+) tri_s (\d+) hits (\d+) lights (\d+)", rs.stdout)
    if rs.returncode == 0 and ms_:
        n, _, _, bc, tc, bs_, ts, hits, lights = [float(v) for v in ms_.groups()]
        per_sample = ((bc + bs_) / 4.0 * 64.0 + (tc + ts) * 48.0 + hits * 144.0 + lights * 96.0) / n + 16.0 / spp
        alg_bytes = per_sample * w * h * spp
        algorithmic = {"bytes_per_sample": round(per_sample, 1), "bytes_per_launch": int(alg_bytes), "traffic_over_algorithmic": round(traffic / alg_bytes, 3),
                       "per_sample": {"box_tests": round((bc + bs_) / n, 2), "tri_tests": round((tc + ts) / n, 2), "shaded_hits": round(hits / n, 3), "light_samples": round(lights / n, 3)},
                       "note": "record bytes the traversal and shading must read per sample (every visit counted, no cache reuse): counter bytes below it mean L2 reuse"}
    return {"workload": "generated stress scene (%d spheres x %d segments): %dx%d %d spp, NEE, colour only" % (spheres, segments, w, h, spp),
            "triangles": int(tri.group(1)) if tri else None, "value": round(w * h * spp / (ms * 1e3), 3), "unit": "Msamples/s", "kernel_ms_avg": round(ms, 3),
            "roofline": {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 5), "traffic": int(traffic),
                         "algorithmic": algorithmic,
                         "definition": "(FETCH_SIZE x 2 + WRITE_SIZE) counter bytes per launch / kernel time: fabric-side traffic, the scene (~70 MB) is Infinity-Cache resident"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--integrator", default="NEE", choices=["NEE", "Pathtrace", "MIS"])
    ap.add_argument("--scene", default="cornell", choices=["cornell", "thinfilm", "ior15", "stress"],
                    help="cornell = BASELINE configs[1] (the headline); thinfilm / ior15 = configs[2] / [3]; stress = generated ~1 M-triangle scene")
    ap.add_argument("--stress-spheres", type=int, default=64)
    ap.add_argument("--stress-segments", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (roofline.frac / traffic become null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary block (1 M-triangle stress scene) of the default line")
    ap.add_argument("--cpu-spp", type=int, default=1024, help="spp of the bounded CPU-baseline sample (the metric's RMSE is quoted at 1024 spp)")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("HJR_CPU_THREADS", "16")),
                    help="oracle threads for the CPU baseline (a 1-GPU box's CPU share is 16 cores)")
    args = ap.parse_args()

    # ---- started plainly with --gpus N > 1: become the launcher.  One rank process per GPU through torch.distributed.run, as a CHILD
    #      process (never an exec of this one), before torch is imported or the GPU is touched; rank 0 of the child prints the JSON line
    #      on the inherited stdout.  Under torchrun (WORLD_SIZE set) this block is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:  # a free rendezvous port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- hardware counters of this workload, collected in child processes BEFORE this process initialises the GPU
    pmc, pmc_full, pmc_error = None, None, None
    config_json = {"cornell": "render_option_c2.json", "thinfilm": "render_option_c3.json", "ior15": "render_option_c4.json"}.get(args.scene)
    if rank == 0 and not args.no_pmc and "HJR_BENCH_DEVICE" not in os.environ:
        try:
            kcfg = config_json
            if args.scene == "stress":
                import subprocess
                import tempfile
                sdir0 = os.path.join(tempfile.gettempdir(), "hjr_stress_%d_%d_r%d" % (args.stress_spheres, args.stress_segments, rank))
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_stress_scene.py"), sdir0, "--spheres",
                                       str(args.stress_spheres), "--segments", str(args.stress_segments)], stdout=subprocess.DEVNULL)
                kcfg = os.path.join(sdir0, "render_option_stress.json")
            lib_path = os.environ.get("HJR_LIB") or os.path.join(ROOT, "henjou-renderer_amd", "libhenjou_hip.so")
            kargs = [lib_path, kcfg, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--reps", "1",
                     "--integrator", str({"NEE": 0, "Pathtrace": 1, "MIS": 2}[args.integrator]), "--rank", "0", "--world", str(world)]
            penv = {"HIP_VISIBLE_DEVICES": str(local_rank), "ROCR_VISIBLE_DEVICES": ""} if world > 1 else {}
            penv = {k: v for k, v in penv.items() if v}
            pmc_full = collect_pmc(kargs + ["--aovs"], env=penv)
            pmc = collect_pmc(kargs, env=penv)
            try:  # instruction classes of the headline launch (one more pass; its absence only drops the `flops` block)
                pmc_full.update(collect_pmc(kargs + ["--aovs"], env=penv, passes=[PMC_FLOP_PASS]))
            except Exception as ex:
                pmc_full["flop_pass_error"] = "%s: %s" % (type(ex).__name__, ex)
        except Exception as ex:  # reported on the line; never silently replaced by a committed file
            pmc_error = "%s: %s" % (type(ex).__name__, ex)
    secondary = None
    if rank == 0 and world == 1 and args.scene == "cornell" and not args.no_pmc and not args.no_secondary and "HJR_BENCH_DEVICE" not in os.environ:
        try:
            secondary = stress_secondary(os.environ.get("HJR_LIB") or os.path.join(ROOT, "henjou-renderer_amd", "libhenjou_hip.so"), args.stress_spheres, args.stress_segments)
        except Exception as ex:
            secondary = {"error": "%s: %s" % (type(ex).__name__, ex)}

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    hjr = entry.load_package()

    if world != args.gpus:  # under a launcher the launcher's world size is the fact
        args.gpus = world
    # rehearsal knobs (never set by the driver): HJR_BENCH_DEVICE pins every rank to one GPU and HJR_BENCH_BACKEND=gloo swaps
    # RCCL for gloo, so that the N > 1 code path can be exercised on a 1-GPU box
    if "HJR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HJR_BENCH_DEVICE"])
    backend = os.environ.get("HJR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    W, H, SPP = args.width, args.height, args.spp
    integ = {"NEE": hjr.INTEGRATOR_NEE, "Pathtrace": hjr.INTEGRATOR_PT, "MIS": hjr.INTEGRATOR_MIS}[args.integrator]

    # ---- scene through the drop-in surface (render_option.json + Model/), resident in HBM before timing
    cwd = os.getcwd()
    os.chdir(hjr.ASSETS)
    try:
        r = hjr.Renderer(local_rank)
        config = config_json
        if args.scene == "stress":
            import subprocess
            import tempfile
            sdir = os.path.join(tempfile.gettempdir(), "hjr_stress_%d_%d_r%d" % (args.stress_spheres, args.stress_segments, rank))
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_stress_scene.py"), sdir, "--spheres",
                                   str(args.stress_spheres), "--segments", str(args.stress_segments)], stdout=subprocess.DEVNULL)
            config = os.path.join(sdir, "render_option_stress.json")
        r.loadRenderOption(config)
        r.render_option.image_width, r.render_option.image_height, r.render_option.max_spp = W, H, SPP
        r.render_option.integrator = integ
        r.loadGLTFfile(r.render_option.gltf_path.decode(), r.render_option.gltf_name.decode())
        r.build()
    finally:
        os.chdir(cwd)
    frame = r.render_option.start_frame
    # N > 1: every rank renders ONLY its 8x8 tiles, packed back to back (HJR_FLAG_PACKED: 1 / N of the frame, no zero fill)
    params, t_frame = r.frame_params(frame, rank=rank, world_size=world, flags=hjr.FLAG_PACKED if world > 1 else 0)
    m, inv = r.scene.transforms(t_frame)
    r.device.set_transforms(m, inv)

    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    fb_albedo = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    fb_normal = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    if world > 1:  # packed AOV buffers: [max owned tiles over the ranks][64][4] (ranks that own one tile fewer leave the last one zero)
        n_max = hjr.owned_tiles(W, H, 0, world)
        pk = torch.zeros((n_max, 64, 4), dtype=torch.float32, device="cuda")
        pk_albedo, pk_normal = torch.zeros_like(pk), torch.zeros_like(pk)
        out_c, out_a, out_n = pk, pk_albedo, pk_normal
    else:
        out_c, out_a, out_n = fb, fb_albedo, fb_normal

    def exchange():
        # the one data-path collective of a frame: RCCL gather of the packed colour tiles onto rank 0 (point-to-point over xGMI, all
        # peers in parallel), scattered into the row-major frame there.  Albedo / normal stay on their ranks: Default mode only writes
        # aov_color to the PNG (renderer.h:1281-1302), as in henjou_cli's multi-GPU path.
        if world > 1:
            hjr.gather_tiles(pk, W, H, device=r.device, dst=0, frame=fb)

    def step_full():  # what the reference's launch produces: aov_color + aov_albedo + aov_normal (renderer.h:1222-1224)
        r.device.render_device(params, out_c.data_ptr(), out_a.data_ptr(), out_n.data_ptr(), stream)
        exchange()

    def step_color():  # colour only (only aov_color reaches the PNG in Default mode)
        r.device.render_device(params, out_c.data_ptr(), None, None, stream)
        exchange()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        for _ in range(args.warmup):
            step()
        fence()
        ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            if world == 1:
                ms.append(r.device.stats()["last_kernel_ms"])  # HIP events recorded on the launch stream around the kernel(s); waits for the step
        fence()
        if world > 1:  # sharded steps are short (1 / N of a frame): no host wait inside the timed region, the last step's kernel time only
            ms.append(r.device.stats()["last_kernel_ms"])
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ms, {0: "persistent megakernel", 1: "workgroup-local wavefront kernel"}[r.device.stats()["pipeline"]]

    elapsed_color, kernel_ms_color, pipe_color = timed(step_color)
    # opt-in approximate-arithmetic kernels (HJR_FLAG_FAST_MATH; megakernel family): a named side block, never the headline
    fast_block = None
    if world == 1:
        params_exact = params
        params, _ = r.frame_params(frame, rank=rank, world_size=world, flags=hjr.FLAG_FAST_MATH)
        elapsed_fast, kernel_ms_fast, _ = timed(step_full)
        fast_frame = fb.clone()
        params = params_exact
    elapsed, kernel_ms, pipe_full = timed(step_full)  # the headline: EXACTLY args.steps steps between the fences
    if world == 1:
        d2 = (fast_frame[..., :3].double() - fb[..., :3].double()) ** 2
        fast_block = {"value": round(float(W) * H * SPP * args.steps / elapsed_fast / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(elapsed_fast / args.steps * 1e3, 3),
                      "kernel_ms_avg": round(sum(kernel_ms_fast) / len(kernel_ms_fast), 3),
                      "rmse_vs_exact_kernel": {"value": float(torch.sqrt(d2.mean()).item()), "spp": SPP, "pixels": W * H},
                      "note": "HJR_FLAG_FAST_MATH, 3 AOVs: hardware reciprocal / sqrt / sin / cos / pow and fused multiply-adds in the shading code (the reference's own "
                              "build is nvcc --use_fast_math); not bit-exact; per-pixel RMSE < 1e-3 at 1024 spp vs the LIBM oracle is tests/test_gpu_fast_math.py"}

    total_samples = float(W) * H * SPP * args.steps
    value = total_samples / elapsed / 1e6
    out = {
        "metric": "Msamples/s at 1920x1080 (path-traced pixel samples per second, whole job)",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "aovs": "color+albedo+normal (the reference raygen's outputs)",
        "color_only": {"value": round(total_samples / elapsed_color / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(elapsed_color / args.steps * 1e3, 3),
                       "kernel_ms_avg": round(sum(kernel_ms_color) / len(kernel_ms_color), 3), "pipeline": pipe_color,
                       "note": "same workload, aov_color only (lean kernel instantiation; Default mode writes only this AOV to the PNG)"},
        "fast_math": fast_block,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (%s, %d triangles; CMJ sample streams from a fixed seed)" % (r.render_option.gltf_name.decode(), r.scene.view.n_triangles),
        "config": {"workload": "%s: %s %dx%d %d spp, %s integrator" % ({"cornell": "BASELINE configs[1]", "thinfilm": "BASELINE configs[2]", "ior15": "BASELINE configs[3]", "stress": "synthetic stress scene"}[args.scene], r.render_option.gltf_name.decode(), W, H, SPP, args.integrator),
                   "scene": args.scene, "triangles": int(r.scene.view.n_triangles),
                   "width": W, "height": H, "spp": SPP, "integrator": args.integrator,
                   "parallelism": "8x8 pixel tiles round-robin over %d GPU(s)%s" % (world, " + RCCL gather of each rank's packed colour tiles onto rank 0" if world > 1 else "")},
    }

    if rank == 0:
        # ---- roofline of the dominant (only) kernel: counters from the counting variant at 1/16 of the samples
        sp = hjr.make_params(W, H, max(SPP // 16, 1), params.camera, frame=params.frame, seed=params.seed, integrator=integ,
                             sky=tuple(params.sky), ibl_intensity=params.ibl_intensity, rank=rank, world_size=world,
                             flags=hjr.FLAG_STATS | (hjr.FLAG_PACKED if world > 1 else 0))
        r.device.render_device(sp, out_c.data_ptr(), None, None, stream)
        torch.cuda.synchronize()
        st = r.device.stats()
        bps = algorithmic_bytes_per_sample(st, SPP)
        samples_per_launch = float(W) * H * SPP / world
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        avg_ms_color = sum(kernel_ms_color) / len(kernel_ms_color)
        algorithmic_gbs = bps * samples_per_launch / (avg_ms * 1e-3) / 1e9

        def counters_view(c, ms):
            """Roofline figures from one set of per-launch counters and the LIVE kernel time of that variant."""
            t = ms * 1e-3
            hbm_bytes = c["FETCH_SIZE"] * 1024.0 * 2.0 + c["WRITE_SIZE"] * 1024.0  # KiB counters; gfx950: FETCH_SIZE reports half of wide reads
            lanes = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1.0)  # average active lanes per VALU wave-instruction
            lane_ops = c["SQ_INSTS_VALU"] * lanes
            clk = c["GRBM_GUI_ACTIVE"] / 8.0  # shader cycles of the profiled launch (sum over the 8 XCDs / 8)
            return {"hbm_bytes": hbm_bytes, "hbm_GBps": hbm_bytes / t / 1e9, "valu_tflops": lane_ops * 2.0 / t / 1e12,
                    "active_lane_frac": lanes / 64.0, "valu_wave_insts": c["SQ_INSTS_VALU"],
                    "simd_cycles_per_valu_inst": clk * 1024.0 / max(c["SQ_INSTS_VALU"], 1.0),
                    "sq_active_inst_any_frac": c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                    "sq_wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)}

        # the dominant kernel under the name rocprofv3 prints for it (template arguments: integrator, counting, block, scene in LDS, 16-bit stack,
        # BVH width, variant 1 = with albedo / normal sums[, fast-math tag]); the other kernels of a step: hjr_finalize_kernel + the tile pre-pass
        lm = st["lds_mode"]
        if pipe_full == "persistent megakernel":
            kname = "void hjr_render_kernel<%d, false, %d, %s, %s, %d, 1, false>(KParams)" % (integ, 1024 if lm in (1, 2) else 256, "true" if lm in (1, 2) else "false", "true" if lm == 2 else "false", 4 if lm == 0 else 2)
        else:
            kname = "void hjr_wavefront_kernel<%d, false, 1024, %s, %s, %d, 1>(KParams)" % (integ, "true" if lm in (1, 2) else "false", "true" if (lm not in (1, 2) or st["stack_lds_entries"] < st["stack_need"]) else "false", 4 if lm == 0 else 2)
        roof = {"kernel": kname, "kernel_note": "%s, 3 AOVs; avg duration of this name in profiles/r03_kernel_stats.csv" % args.integrator, "kernel_ms_avg": round(avg_ms, 3),
                "kernel_Msamples_per_s": round(samples_per_launch / (avg_ms * 1e-3) / 1e6, 3),
                "pipeline": pipe_full,
                "algorithmic": {"bytes_per_sample": round(bps, 1), "GBps": round(algorithmic_gbs, 2),
                                "note": "SURVEY 8d per-visit byte model x measured rate: counts node / triangle reads that LDS and L2 serve, so it is not a roofline",
                                "per_sample": {k: round(st[k] / max(st["samples"], 1), 3) for k in
                                               ("closest_rays", "shadow_rays", "box_tests_closest", "tri_tests_closest",
                                                "box_tests_shadow", "tri_tests_shadow", "shaded_hits", "light_samples")}}}
        if pmc_full is not None:
            v = counters_view(pmc_full, avg_ms)
            hbm_frac = v["hbm_GBps"] / HBM_PEAK_GBS
            if v["valu_tflops"] / VALU_PEAK_TFLOPS >= hbm_frac:  # the roof the launch is closer to: vector ALUs (scene served by LDS / L2) ...
                roof.update({"bound": "valu", "achieved": round(v["valu_tflops"], 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(v["valu_tflops"] / VALU_PEAK_TFLOPS, 5),
                             "definition": "VALU lane-slot utilisation in the peak's units: SQ_INSTS_VALU x (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) active lanes, every "
                                           "instruction counted as one 2-FLOP op, per launch / live kernel time, vs the fp32 vector peak (an upper bound on the useful-FLOP "
                                           "fraction; `flops` counts the fp32 arithmetic instructions only)"})
            else:  # ... or the memory side (scenes beyond the caches; the wavefront kernels' context traffic)
                roof.update({"bound": "hbm", "achieved": round(v["hbm_GBps"], 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 5),
                             "definition": "(FETCH_SIZE x 2 + WRITE_SIZE) counter bytes per launch / live kernel time, vs 8 TB/s (Infinity-Cache hits are counted by FETCH_SIZE)"})
            if "SQ_INSTS_VALU_FMA_F32" in pmc_full:  # fp32 arithmetic only: fma = 2 FLOP, add / mul / transcendental = 1, x active lanes
                lanes = v["active_lane_frac"] * 64.0
                fl = (2.0 * pmc_full["SQ_INSTS_VALU_FMA_F32"] + pmc_full["SQ_INSTS_VALU_ADD_F32"] + pmc_full["SQ_INSTS_VALU_MUL_F32"] + pmc_full["SQ_INSTS_VALU_TRANS_F32"]) * lanes
                roof["flops"] = {"achieved": round(fl / (avg_ms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (avg_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 5),
                                 "wave_insts": {k[14:].lower(): pmc_full[k] for k in PMC_FLOP_PASS if k.startswith("SQ_INSTS_VALU_")},
                                 "salu_wave_insts": pmc_full.get("SQ_INSTS_SALU"), "lds_wave_insts": pmc_full.get("SQ_INSTS_LDS"),
                                 "definition": "(2 x FMA_F32 + ADD_F32 + MUL_F32 + TRANS_F32 wave instructions) x average active lanes / live kernel time"}
            elif "flop_pass_error" in pmc_full:
                roof["flops"] = {"error": pmc_full["flop_pass_error"]}
            roof["traffic"] = int(v["hbm_bytes"])
            roof["counters"] = {"source": "rocprofv3 --pmc passes of tools/kbench inside this run (same library, workload, GPU)",
                                "active_lane_frac": round(v["active_lane_frac"], 4), "valu_wave_insts_per_launch": v["valu_wave_insts"],
                                "simd_cycles_per_valu_inst": round(v["simd_cycles_per_valu_inst"], 3),
                                "hbm_GBps": round(v["hbm_GBps"], 2), "hbm_frac": round(hbm_frac, 6), "valu_TFLOPs": round(v["valu_tflops"], 3),
                                "valu_frac": round(v["valu_tflops"] / VALU_PEAK_TFLOPS, 5),
                                "sq_active_inst_any_frac": round(v["sq_active_inst_any_frac"], 4), "sq_wait_inst_any_frac": round(v["sq_wait_inst_any_frac"], 4)}
            if pmc is not None:
                vc = counters_view(pmc, avg_ms_color)
                roof["counters"]["color_only"] = {"active_lane_frac": round(vc["active_lane_frac"], 4), "valu_TFLOPs": round(vc["valu_tflops"], 3),
                                                  "valu_frac": round(vc["valu_tflops"] / VALU_PEAK_TFLOPS, 5), "hbm_bytes": int(vc["hbm_bytes"])}
        else:
            roof.update({"bound": "valu", "achieved": None, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                         "counters": {"source": "not collected", "reason": pmc_error or ("--no-pmc" if args.no_pmc else "rank / rehearsal run")}})
        out["roofline"] = roof
        if secondary is not None:
            out["secondary"] = {"large_scene": secondary}

        # ---- CPU baseline: the oracle (kind "port": the reference has no CPU path) on a bounded sample of the same
        #      workload: a centred window of the SAME frame (same camera, same resolution) at the metric's 1024 spp.  The same
        #      render doubles as the accuracy check BASELINE.json's metric names: per-pixel RMSE of the product's frame against the
        #      reference arithmetic with glibc transcendentals (LIBM mode) at identical sample streams.
        if world == 1 and not args.no_cpu_baseline:
            import oracle_binding as ob
            cores = os.cpu_count() or 1
            try:
                cores = len(os.sched_getaffinity(0))
            except Exception:
                pass
            cores = max(1, min(cores, args.cpu_threads))
            arrays = r.scene.arrays(t_frame)
            if r.lut is not None:
                arrays["lut_rgba"] = r.lut
            osc = ob.OracleScene(arrays, ob.MATH_LIBM)
            cspp = args.cpu_spp
            # window sized for ~130 Msamples of CPU work (10-30 s on 16 cores)
            ww = max(8, min(W, int(round((130e6 / cspp * W / H) ** 0.5)) // 8 * 8))
            wh = max(8, min(H, int(round(ww * H / W)) // 8 * 8))
            x0, y0 = (W - ww) // 2 // 8 * 8, (H - wh) // 2 // 8 * 8
            rect = (x0, y0, x0 + ww, y0 + wh)
            op = ob.make_params(W, H, cspp, params.camera.as_dict(), frame=params.frame, seed=params.seed, integrator=integ,
                                sky=tuple(params.sky), ibl_intensity=params.ibl_intensity, rect=rect)
            tc = time.perf_counter()
            ocol, _, _, _ = osc.render(op, nthreads=cores, want_aovs=False)
            dtc = time.perf_counter() - tc
            gp = hjr.make_params(W, H, cspp, params.camera, frame=params.frame, seed=params.seed, integrator=integ,
                                 sky=tuple(params.sky), ibl_intensity=params.ibl_intensity)
            r.device.render_device(gp, fb.data_ptr(), None, None, stream)
            torch.cuda.synchronize()
            g = fb[y0:y0 + wh, x0:x0 + ww, :3].cpu().numpy().astype(np.float64)
            o = ocol[y0:y0 + wh, x0:x0 + ww, :3].astype(np.float64)
            rmse = float(np.sqrt(np.mean((g - o) ** 2)))
            out["cpu_baseline"] = {"value": round(ww * wh * cspp / dtc / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                                   "kind": "port",
                                   "sample": "window x %d..%d, y %d..%d of the same %dx%d frame at %d spp = %.2f Msamples, oracle/hjr_oracle.c in LIBM mode, "
                                             "%d pthreads, %.1f s" % (x0, x0 + ww, y0, y0 + wh, W, H, cspp, ww * wh * cspp / 1e6, cores, dtc),
                                   "rmse_gpu_vs_cpu": {"value": rmse, "spp": cspp, "pixels": ww * wh, "tolerance": 1e-3,
                                                       "note": "per-pixel RMSE (linear RGB) of the HIP frame against the CPU restatement with glibc transcendentals, "
                                                               "identical sample streams; no OptiX render can exist here (SURVEY.md section 0)"}}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
