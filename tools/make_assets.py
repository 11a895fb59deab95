#!/usr/bin/env python3
"""Generates the build's own assets (committed under henjou-renderer_amd/assets/):

  LUT/Thin_Film_LUT.png            thin-film interference F0 table.  The reference's LUT image is absent from its repository
                                   (SURVEY.md §0 F5); only the lookup is pinned (disneyBRDF.h:11-14,213-217: u = basecolor.x =
                                   normalised film thickness, v = |cos|).  Contents here: Airy reflectance of a thin film
                                   (n = 1.33 on n = 1.5, thickness 0..1000 nm), spectrally integrated to linear sRGB.
  Model/test_gltf/cornelbox_c2.gltf      BASELINE configs[1] "diffuse/specular only": glass transmission -> 0
  Model/test_gltf/cornelbox_thinfilm.gltf BASELINE configs[2]: materials 0 and 5 carry the custom ThinFilm extension (gltfloader.h:1248-1257)
  Model/test_gltf/cornelbox_ior15.gltf    BASELINE configs[3]: the glass sphere gets KHR_materials_ior 1.5 (gltfloader.h:1226-1235)
  render_option_c3.json / render_option_c4.json
All variants reference the bundled cornelbox.bin.
"""
import copy
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "henjou-renderer_amd", "assets")
GLTF_DIR = os.path.join(ASSETS, "Model", "test_gltf")


def cie_xyz(lam):
    """Wyman/Sloan/Shirley 2013 multi-lobe fit of the CIE 1931 colour matching functions (lam in nm)."""
    def g(x, mu, s1, s2):
        s = np.where(x < mu, s1, s2)
        return np.exp(-0.5 * ((x - mu) / s) ** 2)
    x = 1.056 * g(lam, 599.8, 37.9, 31.0) + 0.362 * g(lam, 442.0, 16.0, 26.7) - 0.065 * g(lam, 501.1, 20.4, 26.2)
    y = 0.821 * g(lam, 568.8, 46.9, 40.5) + 0.286 * g(lam, 530.9, 16.3, 31.1)
    z = 1.217 * g(lam, 437.0, 11.8, 36.0) + 0.681 * g(lam, 459.0, 26.0, 13.8)
    return np.stack([x, y, z], -1)


def thin_film_lut(n=256, n0=1.0, n1=1.33, n2=1.5, dmax_nm=1000.0):
    lam = np.linspace(380.0, 780.0, 81)
    xyz = cie_xyz(lam)
    norm = xyz[:, 1].sum()
    M = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    out = np.zeros((n, n, 4), np.uint8)
    out[..., 3] = 255
    for j in range(n):  # v: cos(theta) at texel centre
        c0 = (j + 0.5) / n
        s0 = np.sqrt(max(0.0, 1 - c0 * c0))
        s1 = n0 * s0 / n1
        c1 = np.sqrt(max(0.0, 1 - s1 * s1))
        s2 = n0 * s0 / n2
        c2 = np.sqrt(max(0.0, 1 - s2 * s2))
        rs01 = (n0 * c0 - n1 * c1) / (n0 * c0 + n1 * c1)
        rp01 = (n1 * c0 - n0 * c1) / (n1 * c0 + n0 * c1)
        rs12 = (n1 * c1 - n2 * c2) / (n1 * c1 + n2 * c2)
        rp12 = (n2 * c1 - n1 * c2) / (n2 * c1 + n1 * c2)
        for i in range(n):  # u: thickness
            d = (i + 0.5) / n * dmax_nm
            phi = 4 * np.pi * n1 * d * c1 / lam
            R = 0
            for r01, r12 in ((rs01, rs12), (rp01, rp12)):
                num = r01 * r01 + r12 * r12 + 2 * r01 * r12 * np.cos(phi)
                den = 1 + r01 * r01 * r12 * r12 + 2 * r01 * r12 * np.cos(phi)
                R = R + 0.5 * num / den
            X = (R[:, None] * xyz).sum(0) / norm
            rgb = np.clip(M @ X, 0.0, 1.0)
            out[j, i, :3] = np.round(rgb * 255.0).astype(np.uint8)
    return out


def main():
    from PIL import Image
    os.makedirs(os.path.join(ASSETS, "LUT"), exist_ok=True)
    lut = thin_film_lut()
    Image.fromarray(lut, "RGBA").save(os.path.join(ASSETS, "LUT", "Thin_Film_LUT.png"), optimize=True)
    base = json.load(open(os.path.join(GLTF_DIR, "cornelbox.gltf")))

    c2 = copy.deepcopy(base)
    c2["materials"][4]["extensions"]["KHR_materials_transmission"]["transmissionFactor"] = 0
    json.dump(c2, open(os.path.join(GLTF_DIR, "cornelbox_c2.gltf"), "w"), indent=1)

    tf = copy.deepcopy(base)
    for i, thickness in ((0, 0.35), (5, 0.62)):
        tf["materials"][i].setdefault("extensions", {})["ThinFilm"] = {"is_ThinFilm": True}
        # basecolor.x doubles as the normalised film thickness (disneyBRDF.h:214)
        tf["materials"][i]["pbrMetallicRoughness"]["baseColorFactor"][0] = thickness
    tf.setdefault("extensionsUsed", []).append("ThinFilm")
    json.dump(tf, open(os.path.join(GLTF_DIR, "cornelbox_thinfilm.gltf"), "w"), indent=1)

    io = copy.deepcopy(base)
    io["materials"][4]["extensions"]["KHR_materials_ior"] = {"ior": 1.5}
    io["extensionsUsed"].append("KHR_materials_ior")
    json.dump(io, open(os.path.join(GLTF_DIR, "cornelbox_ior15.gltf"), "w"), indent=1)

    ro = json.load(open(os.path.join(ASSETS, "render_option_c2.json")))
    for name, gltf, spp in (("c3", "cornelbox_thinfilm.gltf", 1024), ("c4", "cornelbox_ior15.gltf", 1024)):
        r = copy.deepcopy(ro)
        r["Image"]["image_name"] = "cornelbox_" + name
        r["Image"]["max_spp"] = spp
        r["GLTF_file"]["gltf_filename"] = gltf
        json.dump(r, open(os.path.join(ASSETS, "render_option_%s.json" % name), "w"), indent=4)
    c2ro = copy.deepcopy(ro)
    c2ro["GLTF_file"]["gltf_filename"] = "cornelbox_c2.gltf"
    c2ro["Image"]["image_name"] = "cornelbox_c2_diffuse_specular"
    json.dump(c2ro, open(os.path.join(ASSETS, "render_option_c2_nodiel.json"), "w"), indent=4)
    print("assets written; LUT mean", lut[..., :3].mean(axis=(0, 1)))


if __name__ == "__main__":
    main()
