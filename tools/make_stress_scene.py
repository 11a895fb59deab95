#!/usr/bin/env python3
"""Synthetic stress scene for the Henjou hot path: a lit room filled with instanced tessellated spheres so
that the BVH no longer fits in cache and HBM bandwidth is actually exercised.  Deterministic (fixed seed).

    python tools/make_stress_scene.py OUTDIR [--spheres 64] [--segments 128] [--seed 7]

Writes OUTDIR/stress.gltf + stress.bin (+ render_option_stress.json).  One sphere mesh is shared by all sphere nodes; the
reference's loader de-indexes every mesh node separately (gltfloader.h:1354-1512), so triangles = 12 + 2 + spheres * 2*seg*(seg/2 - 1).
"""
import argparse
import json
import os
import struct

import numpy as np


def sphere(seg):
    rings = seg // 2
    v, n, uv = [], [], []
    for j in range(rings + 1):
        th = np.pi * j / rings
        for i in range(seg + 1):
            ph = 2 * np.pi * i / seg
            p = (np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph))
            v.append(p); n.append(p); uv.append((i / seg, j / rings))
    idx = []
    for j in range(rings):
        for i in range(seg):
            a = j * (seg + 1) + i
            b = a + seg + 1
            if j != 0:
                idx += [a, b, a + 1]
            if j != rings - 1:
                idx += [a + 1, b, b + 1]
    return np.array(v, np.float32), np.array(n, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32)


def quad(p0, p1, p2, p3, nrm):
    v = np.array([p0, p1, p2, p3], np.float32)
    n = np.tile(np.array(nrm, np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    return v, n, uv, np.array([0, 1, 2, 0, 2, 3], np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--segments", type=int, default=128)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)
    rng = np.random.default_rng(a.seed)
    blob = bytearray()
    views, accs = [], []

    def add(arr, target=None, typ="VEC3", ct=5126):
        while len(blob) % 4:
            blob.append(0)
        off = len(blob)
        blob.extend(arr.tobytes())
        v = {"buffer": 0, "byteOffset": off, "byteLength": arr.nbytes}
        if target:
            v["target"] = target
        views.append(v)
        acc = {"bufferView": len(views) - 1, "componentType": ct, "count": int(arr.shape[0]), "type": typ}
        if typ == "VEC3" and ct == 5126 and target == 34962:
            acc["min"] = [float(x) for x in arr.min(0)]
            acc["max"] = [float(x) for x in arr.max(0)]
        accs.append(acc)
        return len(accs) - 1

    def prim(geo, material):
        v, n, uv, idx = geo
        return {"attributes": {"POSITION": add(v, 34962), "NORMAL": add(n, 34962), "TEXCOORD_0": add(uv, 34962, "VEC2")},
                "indices": add(idx, 34963, "SCALAR", 5125), "material": material}

    X, Y, Z = 4.0, 2.0, 3.0
    room = [prim(quad((-X, -Y, -Z), (X, -Y, -Z), (X, -Y, Z), (-X, -Y, Z), (0, 1, 0)), 0),   # floor
            prim(quad((-X, Y, -Z), (-X, Y, Z), (X, Y, Z), (X, Y, -Z), (0, -1, 0)), 0),      # ceiling
            prim(quad((-X, -Y, -Z), (-X, Y, -Z), (X, Y, -Z), (X, -Y, -Z), (0, 0, 1)), 1),   # back (green)
            prim(quad((-X, -Y, Z), (X, -Y, Z), (X, Y, Z), (-X, Y, Z), (0, 0, -1)), 2),      # front (red)
            prim(quad((-X, -Y, -Z), (-X, -Y, Z), (-X, Y, Z), (-X, Y, -Z), (1, 0, 0)), 0)]   # far wall; +X is open (camera side)
    light = [prim(quad((-1.5, 0, -1.0), (1.5, 0, -1.0), (1.5, 0, 1.0), (-1.5, 0, 1.0), (0, -1, 0)), 3)]
    sph = sphere(a.segments)
    sphere_meshes = [{"name": "sphere_m%d" % m, "primitives": [prim(sph, m)]} for m in (4, 5, 6, 0)]
    meshes = [{"name": "room", "primitives": room}, {"name": "light", "primitives": light}] + sphere_meshes
    materials = [
        {"name": "white", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1], "metallicFactor": 0, "roughnessFactor": 0.5}},
        {"name": "green", "pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.8, 0.1, 1], "metallicFactor": 0, "roughnessFactor": 0.5}},
        {"name": "red", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.1, 1], "metallicFactor": 0, "roughnessFactor": 0.5}},
        {"name": "emitter", "emissiveFactor": [1, 1, 1], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12}},
         "pbrMetallicRoughness": {"metallicFactor": 0, "roughnessFactor": 0.5}},
        {"name": "glass", "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1}, "KHR_materials_ior": {"ior": 1.5}},
         "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1], "roughnessFactor": 0}},
        {"name": "metal", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.7, 0.3, 1], "metallicFactor": 1, "roughnessFactor": 0.3}},
        {"name": "plastic", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.3, 0.8, 1], "metallicFactor": 0, "roughnessFactor": 0.2}},
    ]
    nodes = [{"name": "Camera", "camera": 0, "translation": [9.5, 0.0, 0.0], "rotation": [0, 0.7071068286895752, 0, 0.7071068286895752]},
             {"name": "room", "mesh": 0},
             {"name": "light", "mesh": 1, "translation": [0, Y - 0.01, 0]}]
    for s in range(a.spheres):
        r = float(rng.uniform(0.18, 0.45))
        pos = [float(rng.uniform(-X + 0.6, X - 0.6)), float(rng.uniform(-Y + 0.5, Y - 0.7)), float(rng.uniform(-Z + 0.6, Z - 0.6))]
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        nodes.append({"name": "s%d" % s, "mesh": 2 + (s % 4), "translation": pos, "scale": [r, r, r], "rotation": [float(x) for x in q]})
    g = {"asset": {"version": "2.0", "generator": "henjou tools/make_stress_scene.py"},
         "extensionsUsed": ["KHR_materials_emissive_strength", "KHR_materials_transmission", "KHR_materials_ior"],
         "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes,
         "cameras": [{"type": "perspective", "perspective": {"yfov": 0.60, "aspectRatio": 1.7777, "znear": 0.1, "zfar": 100}}],
         "materials": materials, "meshes": meshes, "accessors": accs, "bufferViews": views,
         "buffers": [{"byteLength": len(blob), "uri": "stress.bin"}]}
    open(os.path.join(a.outdir, "stress.bin"), "wb").write(bytes(blob))
    json.dump(g, open(os.path.join(a.outdir, "stress.gltf"), "w"))
    tris = 12 - 2 + 2 + a.spheres * (len(sph[3]) // 3)
    ro = {"Image": {"image_width": a.width, "image_height": a.height, "image_name": "stress", "image_directory": "./", "max_spp": a.spp},
          "Render_mode": "Default", "GLTF_file": {"gltf_filepath": os.path.abspath(a.outdir) + "/", "gltf_filename": "stress.gltf"},
          "Camera": {"allow_camera_animation": True, "camera_position": [9.5, 0, 0], "camera_direction": [-1, 0, 0], "camera_fov": 35.0},
          "PTX_File": {"ptxfile_path": "unused"}, "Animation": {"fps": 24, "start_frame": 1, "end_frame": 2, "time_limit": 5.0},
          "Sky": {"IBL_path": "none", "IBL_intensity": 1.0, "use_IBL": False, "scene_sky_default": [0.5, 0.6, 0.8]},
          "Option": {"use_date": False, "save_renderOption": False}, "LUT": {"LUT_path": "./LUT/Thin_Film_LUT.png"}}
    json.dump(ro, open(os.path.join(a.outdir, "render_option_stress.json"), "w"), indent=1)
    print("stress scene: %d spheres x %d triangles = %d triangles, %.1f MB" % (a.spheres, len(sph[3]) // 3, tris, len(blob) / 1e6))


if __name__ == "__main__":
    main()
