#!/usr/bin/env python3
"""Per-pixel comparison of two renders (PFM float32 or binary PPM 8-bit), e.g. a frame of this renderer against a
screenshot of the Vulkan original: max-abs difference, share of pixels within a tolerance, share identical, and an
optional difference image.   python tools/image_diff.py a.pfm b.pfm [--tol 1e-3] [--out diff.ppm]"""
import argparse
import sys

import numpy as np


def read_image(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic in (b"PF", b"PF4"):
            w, h = map(int, f.readline().split())
            scale = float(f.readline())
            ch = 3 if magic == b"PF" else 4
            a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, ch)[::-1]
            return a[..., :3].astype(np.float32)
        if magic == b"P6":
            tok = []
            while len(tok) < 3:
                line = f.readline()
                if not line.startswith(b"#"):
                    tok += line.split()
            w, h, mx = map(int, tok[:3])
            a = np.frombuffer(f.read(w * h * 3), np.uint8).reshape(h, w, 3)
            return a.astype(np.float32) / mx
    raise SystemExit("unsupported image format: %s" % path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--clamp8", action="store_true", help="compare the clamped 8-bit views (what the reference's UNORM image holds)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    A, B = read_image(args.a), read_image(args.b)
    if A.shape != B.shape:
        raise SystemExit("size mismatch %s vs %s" % (A.shape, B.shape))
    if args.clamp8:
        A, B = np.round(np.clip(A, 0, 1) * 255) / 255, np.round(np.clip(B, 0, 1) * 255) / 255
    d = np.abs(A - B).max(axis=2)
    print("pixels %d  max-abs %.6g  within %.3g: %.4f %%  identical: %.4f %%" % (d.size, d.max(), args.tol, 100 * (d <= args.tol).mean(), 100 * (d == 0).mean()))
    if args.out:
        img = (np.clip(d / max(d.max(), 1e-12), 0, 1) * 255).astype(np.uint8)
        with open(args.out, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(np.repeat(img[..., None], 3, axis=2).tobytes())
    return 0 if (d <= args.tol).mean() >= 0.999 else 1


if __name__ == "__main__":
    sys.exit(main())
