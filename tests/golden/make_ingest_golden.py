#!/usr/bin/env python3
"""Golden fixtures for the ingest rows (a1-a4, a20) from the reference's own vendored loaders.

Runs oracle/_ref/ref_ingest (oracle/ref_ingest.cpp compiled in place against
/root/reference/include/tiny_obj_loader.h and stb_image.h) on the resource files and records sizes,
SHA-256 of the raw outputs and a few leading values.  Needs the authoring container
(/root/reference present); the fixture it writes is what travels.

    make -C oracle ref && python tests/golden/make_ingest_golden.py
"""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_ingest")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    out = {"obj": {}, "jpg": {}}
    with tempfile.TemporaryDirectory() as td:
        for name in ("cube", "cube_scene", "teapot"):
            pre = os.path.join(td, name)
            subprocess.check_call([REF_BIN, "obj", os.path.join(ROOT, "resources", name + ".obj"), pre], stdout=subprocess.DEVNULL)
            rec = {}
            for ext, dt in (("vertices.f32", np.float32), ("normals.f32", np.float32), ("vidx.u32", np.uint32), ("nidx.i32", np.int32), ("faces.u32", np.uint32)):
                raw = open(pre + "." + ext, "rb").read()
                arr = np.frombuffer(raw, dtype=dt)
                rec[ext] = {"count": int(arr.size), "sha256": sha(raw), "head": [float(x) if dt == np.float32 else int(x) for x in arr[:9]]}
            out["obj"][name] = rec
        for sky in ("skybox_texture_test", "skybox_texture_sea"):
            for face in ("right", "left", "top", "bottom", "front", "back"):  # src/main.cpp:2064-2071 order
                dst = os.path.join(td, "f.rgba")
                whc = subprocess.check_output([REF_BIN, "jpg", os.path.join(ROOT, "resources", sky, face + ".jpg"), dst]).split()
                raw = open(dst, "rb").read()
                a = np.frombuffer(raw, np.uint8).reshape(-1, 4)
                out["jpg"][sky + "/" + face] = {"w": int(whc[0]), "h": int(whc[1]), "channels_in_file": int(whc[2]), "sha256": sha(raw),
                                                "mean_rgb": [float(a[:, k].mean()) for k in range(3)], "first_px": [int(x) for x in a[0]],
                                                "center_px": [int(x) for x in a[(int(whc[1]) // 2) * int(whc[0]) + int(whc[0]) // 2]]}
    dst = os.path.join(ROOT, "tests", "golden", "ingest_golden.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
