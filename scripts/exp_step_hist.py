"""Steps per ray of k_extend (node and leaf visits, both levels) and what a wave pays for them (library built with
-DRT_EXP=1024).  Prints the histogram's summary, the measured sum over waves of the longest walk, and a simulation of
"cut at C steps, continue the unfinished rays in fresh waves" from the histogram (waves filled with random rays).
usage: RT_MI355X_LIB=<variant .so> python scripts/exp_step_hist.py [scene] [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402,F401

import raytracer_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "knight"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if name == "knight":
    import gen_knight

    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
r = rt.Renderer(0, max_paths=1 << 26)
r.upload(sc)
W = 600
H = sc.image_height(W)
lib = rt.load_library()
hist = (C.c_ulonglong * 2048)()
waves = (C.c_ulonglong * 2)()
lib.rt_exp_step_hist(hist, waves)
lib.rt_exp_cut_hist((C.c_ulonglong * 5)(), (C.c_ulonglong * (5 * 2048))())
mesh_live = (C.c_ulonglong * 65)()
lib.rt_exp_mesh_live(mesh_live)
os.environ.setdefault("RT_TAIL", "0")  # every bounce through k_extend: the tail launch would add its own walks to the counts
r.render_rows(W, H, 0, H, spp, 50, seed=0, want_sums=False)
lib.rt_exp_step_hist(hist, waves)
h = np.array(list(hist), dtype=np.int64)
n = h.sum()
steps = np.arange(2048)
print(f"{name}: {n} rays in k_extend, {(h * steps).sum() / n:.2f} steps per ray, longest {np.nonzero(h)[0].max()}")
cum = np.cumsum(h) / n
for q in (0.5, 0.9, 0.99, 0.999):
    print(f"  {q * 100:5.1f} % of the rays need <= {np.searchsorted(cum, q)} steps")
print(f"  measured: {waves[1]} waves, longest walk of a wave {waves[0] / waves[1]:.1f} steps on average "
      f"(lanes busy {(h * steps).sum() / (64.0 * waves[0]):.3f})")
first = (C.c_ulonglong * 5)()
left = (C.c_ulonglong * (5 * 2048))()
lib.rt_exp_cut_hist(first, left)
rng = np.random.default_rng(0)
L = np.array(list(left), dtype=np.int64).reshape(5, 2048)
uncut = float(waves[0])
print(f"  the waves as they are: {uncut / waves[1]:.1f} wave-steps per wave; all lanes busy would be {(h * steps).sum() / 64.0 / waves[1]:.1f}")
for c, cut in enumerate((8, 16, 24, 32, 48)):
    total = float(first[c])
    rest = L[c]
    nleft = int(rest.sum())
    line = f"  cut at {cut:2d}: first pass {total / uncut * 100:5.1f} % of uncut, {nleft / n * 100:5.2f} % of the rays left"
    # second pass: the rays that are left in fresh waves of 64 (random order), run to the end; or cut again at 4 x
    if nleft >= 64:
        sample = rng.choice(steps, size=64 * min(200000, max(1, nleft // 64)), p=rest / nleft).reshape(-1, 64)
        to_end = sample.max(axis=1).mean() * (nleft / 64.0)
        again = np.minimum(sample, 4 * cut).max(axis=1).mean() * (nleft / 64.0)
        rest2 = sample[sample > 4 * cut] - 4 * cut
        frac2 = len(rest2) / sample.size
        if len(rest2) >= 64:
            s2 = rng.choice(rest2, size=(max(1, len(rest2) // 64), 64))
            again += s2.max(axis=1).mean() * (nleft * frac2 / 64.0)
        line += f"; second pass to the end: total {(total + to_end) / uncut * 100:5.1f} %; three passes (cut at {4 * cut}): {(total + again) / uncut * 100:5.1f} %"
    print(line)

lib.rt_exp_mesh_live(mesh_live)
ml = np.array(list(mesh_live), dtype=np.int64)
if ml.sum():
    print(f"  node steps of the one-lane mesh walk: {ml.sum()} wave-steps, {(ml * np.arange(65)).sum() / ml.sum():.1f} lanes in them on average")
    c = np.cumsum(ml) / ml.sum()
    for k in (1, 2, 4, 8, 16, 32):
        print(f"    with <= {k:2d} lanes still walking: {c[k] * 100:5.1f} % of the wave-steps")
