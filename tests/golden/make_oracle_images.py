#!/usr/bin/env python3
"""Regression goldens of the CPU oracle itself: SHA-256 of a few small frames it renders (scene, size and
uniforms listed below), plus the ray counts.  These do NOT pin the oracle to the reference (nothing can: the
reference has no image fixtures and cannot run here) — they pin it to its own past, so that an
accidental change of the canonical arithmetic shows up in the CPU suite and not only as a GPU/oracle mismatch.
    python tests/golden/make_oracle_images.py        # rewrites tests/golden/oracle_images.json
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import scenes  # noqa: E402
from vulkan_raytracing_amd import host  # noqa: E402

RES = scenes.RES


def cases():
    """name -> (ScenePair, W, H); also imported by tests/test_oracle.py"""
    out = {}
    inst = [host.make_instance(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), 0, 0)]
    u = host.default_uniforms(max_bounce_count=0, samples_per_pixel=1, center_object_type=0, orbiting_object_type=0)
    out["cfg1_cube_scene_96x96_depth1_spp1"] = (scenes.ScenePair([os.path.join(RES, "cube_scene.obj")], inst, u, sky=scenes.synthetic_skybox(64)), 96, 96)
    for name, (ct, ot, mb) in {"teapot_mirror_cube_diffuse_depth2": (1, 0, 1), "teapot_glass_cube_mirror_depth6": (2, 1, 5)}.items():
        sp = scenes.two_object_scene(os.path.join(RES, "teapot.obj"), os.path.join(RES, "cube.obj"), ct, ot, mb, 2,
                                     sky=scenes.synthetic_skybox(64), time_param=0.4)
        out[name + "_120x68_spp2"] = (sp, 120, 68)
    return out


def render_record(sp, W, H):
    img, rc = sp.orc.render(W, H)
    img = np.ascontiguousarray(img, np.float32)
    return {"sha256": hashlib.sha256(img.tobytes()).hexdigest(), "rays": [int(rc[0]), int(rc[1]), int(rc[2])],
            "mean_rgb": [float(np.float64(img[..., k].mean())) for k in range(3)]}


if __name__ == "__main__":
    rec = {name: dict(render_record(sp, W, H), width=W, height=H) for name, (sp, W, H) in cases().items()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_images.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))
