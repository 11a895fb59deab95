"""Independent numpy restatement of the reference's glTF -> SceneData rules (loader/gltfloader.h:1068-1601), used only to
cross-check the C++ loader in libhenjou_hip.so.  Handles .gltf with external .bin buffers."""
import json
import os

import numpy as np

CT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def load(dirname, filename, allow_camera_animation=True):
    g = json.load(open(os.path.join(dirname, filename)))
    bufs = [open(os.path.join(dirname, b["uri"]), "rb").read() for b in g["buffers"]]

    def acc(i):
        a = g["accessors"][i]
        v = g["bufferViews"][a["bufferView"]]
        dt = np.dtype(CT[a["componentType"]])
        nc = NC[a["type"]]
        stride = v.get("byteStride", 0) or dt.itemsize * nc
        off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        raw = np.frombuffer(bufs[v["buffer"]], dtype=np.uint8)
        out = np.zeros((a["count"], nc), dtype=dt)
        for k in range(a["count"]):
            out[k] = np.frombuffer(raw[off + k * stride: off + k * stride + dt.itemsize * nc].tobytes(), dtype=dt)
        return out

    mats = []
    for m in g.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        base = pbr.get("baseColorFactor", [1, 1, 1, 1])
        em = np.array(m.get("emissiveFactor", [0, 0, 0]), dtype=np.float32)
        d = dict(basecolor=np.array(base[:3], dtype=np.float32), roughness=np.float32(pbr.get("roughnessFactor", 1.0)),
                 metallic=np.float32(pbr.get("metallicFactor", 1.0)), is_light=int(float(em[0] + em[1] + em[2]) > 0.0),
                 sheen=np.float32(0), clearcoat=np.float32(0), transmission=np.float32(0), ior=np.float32(1), is_thinfilm=0)
        ext = m.get("extensions", {})
        if "KHR_materials_clearcoat" in ext and "clearcoatFactor" in ext["KHR_materials_clearcoat"]:
            d["clearcoat"] = np.float32(ext["KHR_materials_clearcoat"]["clearcoatFactor"])
        if "KHR_materials_sheen" in ext and "sheenRoughnessFactor" in ext["KHR_materials_sheen"]:
            d["sheen"] = np.float32(ext["KHR_materials_sheen"]["sheenRoughnessFactor"])
        if "KHR_materials_transmission" in ext and "transmissionFactor" in ext["KHR_materials_transmission"]:
            d["transmission"] = np.float32(ext["KHR_materials_transmission"]["transmissionFactor"])
        if "KHR_materials_ior" in ext and "ior" in ext["KHR_materials_ior"]:
            d["ior"] = np.float32(ext["KHR_materials_ior"]["ior"])
        if "KHR_materials_emissive_strength" in ext and "emissiveStrength" in ext["KHR_materials_emissive_strength"]:
            em = (em * np.float32(ext["KHR_materials_emissive_strength"]["emissiveStrength"])).astype(np.float32)
        if "ThinFilm" in ext and "is_ThinFilm" in ext["ThinFilm"]:
            d["is_thinfilm"] = 1
        d["emission"] = em
        d["ideal_specular"] = int(d["roughness"] == 0 and d["transmission"] > 0)
        mats.append(d)

    V, N, T, MID, PO, LP, LE, INST_ANIM = [], [], [], [], [], [], [], []
    cam = None
    for ni, node in enumerate(g.get("nodes", [])):
        if "mesh" in node:
            PO.append(len(MID))
            for prim in g["meshes"][node["mesh"]]["primitives"]:
                idx = acc(prim["indices"]).reshape(-1).astype(np.int64)
                pos = acc(prim["attributes"]["POSITION"])
                nor = acc(prim["attributes"]["NORMAL"]) if "NORMAL" in prim["attributes"] else None
                tex = acc(prim["attributes"]["TEXCOORD_0"]) if "TEXCOORD_0" in prim["attributes"] else None
                for t in range(len(idx) // 3):
                    tri = idx[3 * t: 3 * t + 3]
                    v = pos[tri]
                    if nor is not None:
                        n = nor[tri]
                    else:
                        c = np.cross((v[1] - v[0]).astype(np.float32), (v[2] - v[0]).astype(np.float32)).astype(np.float32)
                        c = c * (np.float32(1) / np.sqrt(np.float32(np.dot(c, c))))
                        n = np.stack([c, c, c])
                    uv = tex[tri] if tex is not None else np.zeros((3, 2), np.float32)
                    V.append(v); N.append(n); T.append(uv)
                    if mats[prim["material"]]["is_light"]:
                        LP.append(len(MID))
                        LE.append(mats[prim["material"]]["emission"])
                    MID.append(prim["material"])
            INST_ANIM.append(ni)
        elif "camera" in node and allow_camera_animation:
            cam = dict(animation_id=ni, fov=np.float32(g["cameras"][node["camera"]]["perspective"]["yfov"]))
    return dict(vertices=np.array(V, np.float32).reshape(-1, 3), normals=np.array(N, np.float32).reshape(-1, 3),
                texcoords=np.array(T, np.float32).reshape(-1, 2), material_ids=np.array(MID, np.uint32),
                prim_offsets=np.array(PO, np.uint32), light_prim_ids=np.array(LP, np.uint32),
                light_prim_emission=np.array(LE, np.float32).reshape(-1, 3), instance_animation_id=np.array(INST_ANIM, np.uint32),
                materials=mats, camera=cam, gltf=g)
