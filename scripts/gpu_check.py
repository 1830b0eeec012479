"""Ad-hoc GPU check: trace + render parity vs the oracle for the example scenes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from oracle import binding as orc

names = sys.argv[1:] or ["cornell", "two_spheres", "globe", "blocks_ball", "blue_ball"]
for name in names:
    path = os.path.join(ROOT, "examples", name + ".json5")
    scene = rt.Scene.load(path, ROOT)
    oracle = orc.OracleScene.load(path, ROOT)
    print(name, scene.counts(), flush=True)
    W = 96
    H = scene.image_height(W)
    r = rt.Renderer(0)
    r.upload(scene)
    rays = r.primary_rays(W, H, 0, H, 0, 11)
    orays = oracle.primary_rays(W, H, 0, H, 0, 11)
    print("  raygen max |d diff|", np.abs(rays["d"] - orays["d"]).max(), "o", np.abs(rays["o"] - orays["o"]).max(), "time", np.abs(rays["time"] - orays["time"]).max())
    got = r.trace(orays, 11)
    want = oracle.trace(orays, 11)
    same_hit = (got["hit"] == want["hit"])
    same_id = same_hit & (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    h = want["hit"] == 1
    rel = np.abs(got["t"][h & same_id] - want["t"][h & same_id]) / np.maximum(np.abs(want["t"][h & same_id]), 1e-300)
    print(f"  trace: n={len(want)} hits={h.sum()} hit-mismatch={(~same_hit).sum()} id-mismatch={(~same_id).sum()} max rel t err={rel.max() if rel.size else 0:.3e}",
          "n err", np.abs(got["n"][h & same_id] - want["n"][h & same_id]).max() if h.any() else 0,
          "uv err", max(np.abs(got["u"][h&same_id] - want["u"][h&same_id]).max(), np.abs(got["v"][h&same_id] - want["v"][h&same_id]).max()) if h.any() else 0,
          "mat/front mismatch", ((got["material"] != want["material"]) | (got["front_face"] != want["front_face"]))[h & same_id].sum(), flush=True)
    t0 = time.time()
    rgb8, sums, stats = r.render_rows(W, H, 0, H, 8, 50, 11)
    t1 = time.time()
    o8, osums, orays_n = oracle.render(W, H, 0, H, 8, 50, 11, threads=8)
    t2 = time.time()
    close = np.isclose(sums, osums, rtol=1e-6, atol=1e-9).all(axis=2)
    print(f"  render: gpu rays {stats.rays} oracle rays {orays_n} pixels equal {close.mean()*100:.2f}% u8 equal {(rgb8==o8).mean()*100:.2f}% "
          f"gpu {stats.ms_total:.1f} ms (ext {stats.ms_extend:.1f} shade {stats.ms_shade:.1f}) wall {t1-t0:.2f}s oracle {t2-t1:.2f}s mean gpu {sums.mean():.5f} oracle {osums.mean():.5f}", flush=True)
