#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ with the CPU oracle.

The reference is Rust and cannot be built or run here, so these
vectors come from oracle/ (the C++ restatement), not from the reference binary; they pin the
oracle against regressions and give the GPU tests inputs that do not need the oracle at
all.  Per scene:
  <scene>_rays.npz   rays (rt_ray layout) + the oracle's hit records for them:
                     a 40x40 camera sweep plus secondary rays leaving the primary hit points
                     in random directions, and a few rays with custom [tmin, tmax]
  <scene>_image.npz  per-pixel radiance sums of a 40-pixel-wide render, 32 spp, 12 bounces
  traversal_stats.json  mean nodes / primitives / leaves / transforms per ray (bench roofline)

usage: python scripts/make_golden.py            (run from the repo root, after `make`)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import binding as orc  # noqa: E402
import gen_knight  # noqa: E402

SCENES = ["two_spheres", "cornell", "globe", "blocks_ball", "knight", "blue_ball"]
OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20260101


def rays_for(scene, rng):
    W = 40
    H = scene.image_height(W)
    prim = scene.primary_rays(W, H, 0, H, sample=0, seed=SEED)
    hits = scene.trace(prim, seed=SEED)
    sec = np.zeros(0, dtype=orc.RAY_DTYPE)
    ok = hits["hit"] == 1
    if ok.any():
        n = int(ok.sum())
        sec = np.zeros(n, dtype=orc.RAY_DTYPE)
        sec["o"] = hits["p"][ok]
        d = rng.normal(size=(n, 3))
        sec["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        sec["time"] = rng.uniform(0, 1, size=n)
        sec["tmin"] = 0.001
        sec["tmax"] = np.inf
    # custom bounds: reuse primary rays with a window around / before / after the hit
    win = prim[: min(200, len(prim))].copy()
    t = np.where(hits["hit"][: len(win)] == 1, hits["t"][: len(win)], 10.0)
    third = len(win) // 3
    win["tmax"][:third] = t[:third] * 0.5            # bound ends before the nearest hit
    win["tmin"][third:2 * third] = t[third:2 * third] * 1.0001  # bound starts just behind it
    win["tmin"][2 * third:] = -1.0e9                 # negative tmin: hits behind the origin count
    return np.concatenate([prim, sec, win])


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
    stats = {}
    for name in SCENES:
        scene = orc.OracleScene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
        rng = np.random.default_rng(sum(map(ord, name)))
        rays = rays_for(scene, rng)
        hits = scene.trace(rays, seed=SEED)
        np.savez_compressed(os.path.join(OUT, name + "_rays.npz"), rays=rays, hits=hits, seed=np.uint64(SEED))
        W = 40
        H = scene.image_height(W)
        rgb8, sums, nrays = scene.render(W, H, 0, H, 32, 12, seed=SEED, threads=8)
        np.savez_compressed(os.path.join(OUT, name + "_image.npz"), sums=sums, rgb8=rgb8, rays=np.uint64(nrays),
                            width=W, height=H, spp=32, max_bounces=12, seed=np.uint64(SEED))
        W = 600
        H = scene.image_height(W)
        st = scene.traversal_stats(W, H, 1, 50, seed=0, pixel_stride=3)
        st["rays_per_path"] = st["rays"] / (len(range(0, W, 3)) * len(range(0, H, 3)))
        stats[name] = st
        print(name, len(rays), "rays,", int((hits["hit"] == 1).sum()), "hits; image", sums.shape, st)
    with open(os.path.join(OUT, "traversal_stats.json"), "w") as f:
        json.dump({"note": "oracle counters (scripts/make_golden.py): 600 px wide, 1 spp on every 3rd pixel, 50 bounces, seed 0",
                   "scenes": stats}, f, indent=1)


if __name__ == "__main__":
    main()
