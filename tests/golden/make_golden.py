#!/usr/bin/env python3
"""Generates the committed golden images from the CPU oracle (not from the reference: it cannot be built or run here,
DESIGN.md §3).  Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402
from scene_util import Cornell  # noqa: E402

c = Cornell()
for name, mode in (("portable", ob.MATH_PORTABLE), ("libm", ob.MATH_LIBM)):
    img, _, _, _ = ob.OracleScene(c.arrays, mode).render(c.oracle_params(64, 64, 8), want_aovs=False)
    np.save(os.path.join(HERE, "cornelbox_64x64_8spp_nee_%s.npy" % name), img)
    print(name, img[..., :3].mean())
