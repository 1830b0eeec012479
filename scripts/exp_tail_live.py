"""Histogram of live lanes per wave and iteration inside the tail launch (library built with -DRT_EXP=256).
usage: RT_MI355X_LIB=<variant .so> python scripts/exp_tail_live.py [scene] [spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402,F401

import raytracer_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "knight"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 13
if name == "knight":
    import gen_knight

    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
r = rt.Renderer(0, max_paths=1 << 26)
r.upload(sc)
W = 600
H = sc.image_height(W)
lib = rt.load_library()
out = (C.c_ulonglong * 65)()
lib.rt_exp_live(out, 1)
clocks = (C.c_ulonglong * (3 * 8192))()
lib.rt_exp_wave_clock(clocks)
ticks = (C.c_ulonglong * 65)()
lib.rt_exp_live_ticks(ticks)
r.render_rows(W, H, 0, H, spp, 50, seed=0, want_sums=False)
lib.rt_exp_live(out, 1)
h = list(out)
total = sum(h)
print(f"{name} {spp} spp: {total} wave-iterations in the tail launch")
cum = 0
for k in range(1, 65):
    cum += h[k]
    if k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        print(f"  <= {k:2d} live lanes: {cum / total * 100:5.1f} % of the wave-iterations")
lib.rt_exp_live_ticks(ticks)
tk = list(ticks)
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 63), (64, 64)):
    n = sum(h[lo:hi + 1])
    if n:
        print(f"  iterations with {lo:2d}..{hi:2d} live lanes: {sum(tk[lo:hi + 1]) / n / 100.0:7.1f} us each ({n} of them)")

# when the waves of the (last) tail launch started, saw the queue run dry, and ended
lib.rt_exp_wave_clock(clocks)
import numpy as np  # noqa: E402

c = np.array(list(clocks), dtype=np.int64).reshape(3, 8192)
used = c[2] > 0
t0 = c[0][used].min()
start, dry, end = [(c[k][used] - t0) / 100.0 for k in range(3)]  # us
print(f"  {used.sum()} waves; launch {end.max():.0f} us; waves start by {np.percentile(start, 99):.0f} us (99 %)")
print(f"  queue dry for the first wave at {dry[dry > 0].min():.0f} us, for the last at {dry.max():.0f} us")
for q in (10, 50, 90, 99, 99.9, 100):
    print(f"  {q:5.1f} % of the waves have ended by {np.percentile(end, q):7.0f} us")
