"""Measures what the image parity tests assert (run on the GPU box): per scene the fraction of pixels whose f64 sums equal the
oracle's (rtol 1e-6), how far the others are, the oracle's own two-seed noise, and the ray-count difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_knight  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from oracle import binding as orc  # noqa: E402

gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
LUM = np.array([0.2126, 0.7152, 0.0722])
for name in ["two_spheres", "cornell", "globe", "blocks_ball", "knight", "blue_ball"]:
    path = os.path.join(ROOT, "examples", name + ".json5")
    sc = rt.Scene.load(path, ROOT)
    r = rt.Renderer(0)
    r.upload(sc)
    o = orc.OracleScene.load(path, ROOT)
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    H = sc.image_height(W)
    spp, mb = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 50
    _, sums, st = r.render_rows(W, H, 0, H, spp, mb, seed=2024)
    _, osums, orays = o.render(W, H, 0, H, spp, mb, seed=2024, threads=16)
    _, osums2, _ = o.render(W, H, 0, H, spp, mb, seed=2025, threads=16)
    close = np.isclose(sums, osums, rtol=1e-6, atol=1e-9).all(axis=2)
    dl = np.abs((sums - osums) @ LUM) / spp
    noise = np.abs((osums - osums2) @ LUM) / spp
    bad = ~close
    print(f"{name:12s} {W}x{H}: equal {close.mean():.5f} ({bad.sum()} pixels differ), max |dL| of those {dl[bad].max() if bad.any() else 0:.4g}, "
          f"two-seed noise: median {np.median(noise):.4g} p99 {np.quantile(noise, 0.99):.4g} max {noise.max():.4g}; rays {st.rays} vs {orays} "
          f"(diff {int(st.rays) - orays}); max rel err of equal pixels {np.max(np.abs(sums - osums)[close] / (np.abs(osums)[close] + 1e-9)):.3g}", flush=True)
