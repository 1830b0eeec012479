"""Experiment: how much faster is k_extend on incoherent secondary rays when they are sorted?
Run under rocprofv3 --kernel-trace; compare the k_extend durations of the three rt_trace calls."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_knight
import raytracer_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "knight"
gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
r = rt.Renderer(device=0, precision=0); r.upload(sc)
W = 1600; H = sc.image_height(W)
prim = r.primary_rays(W, H, 0, H, 0, seed=1)
hits = r.trace(prim, seed=1)
ok = hits["hit"] != 0
rng = np.random.default_rng(0)
n = hits["n"][ok]; p = hits["p"][ok]
d = rng.normal(size=n.shape); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = n + d; d /= np.linalg.norm(d, axis=1, keepdims=True)
sec = np.zeros(len(p), dtype=rt.RAY_DTYPE)
sec["o"] = p; sec["d"] = d; sec["tmin"] = 0.001; sec["tmax"] = np.inf
print("secondary rays", len(sec))
r.trace(sec, seed=2)                                   # 1: pixel order (what the renderer has)
octant = (d[:, 0] > 0) * 4 + (d[:, 1] > 0) * 2 + (d[:, 2] > 0) * 1
tile = np.arange(len(sec)) // 4096
order = np.lexsort((octant, tile))                     # 2: octant-sorted inside each 4096-ray tile
r.trace(sec[order], seed=2)
lo = p.min(axis=0); hi = p.max(axis=0)
q = np.clip(((p - lo) / (hi - lo + 1e-12) * 63).astype(np.int64), 0, 63)
cell = (q[:, 0] << 12) | (q[:, 1] << 6) | q[:, 2]
order = np.lexsort((cell, octant))                     # 3: globally by octant, then origin cell
r.trace(sec[order], seed=2)
perm = rng.permutation(len(sec))
r.trace(sec[perm], seed=2)                             # 4: fully shuffled
