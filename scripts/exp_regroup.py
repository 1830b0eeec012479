"""What would regrouping the rays of a k_extend launch buy?  Records the steps (node and leaf visits) of every ray of the
launch of one bounce by its place in the launch (library built with -DRT_EXP=1024) and sums the longest walk per wave
for the order as it is, for the rays of each 256-ray block (or 4096-ray tile) split into "walks the mesh" / "does not",
and for a full sort by length inside the block / tile (the bound of what any regrouping by a predicted length can reach).
usage: RT_MI355X_LIB=<variant .so> python scripts/exp_regroup.py [scene] [spp]"""
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
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 17
if name == "knight":
    import gen_knight

    gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
os.environ["RT_TAIL"] = "0"
os.environ["RT_LANES"] = "1"
sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
r = rt.Renderer(0, max_paths=1 << 26)
r.upload(sc)
W = 600
H = sc.image_height(W)
lib = rt.load_library()
lib.rt_exp_ray_steps.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
N = 1 << 23
for depth in (0, 1, 2, 4, 8):
    lib.rt_exp_ray_steps(None, 0, depth)
    _, _, st = r.render_rows(W, H, 0, H, spp, 50, seed=0, want_sums=False)
    buf = np.zeros(N, dtype=np.uint16)
    lib.rt_exp_ray_steps(buf.ctypes.data, N, 0xffffffff)
    n = int(np.max(np.nonzero(buf)[0])) + 1 if buf.any() else 0
    n -= n % 4096
    if n == 0:
        continue
    octant = (buf[:n] >> 12).astype(np.int64)
    s = (buf[:n] & 4095).astype(np.int64)

    def cost(order_fn, group):
        g = s.reshape(-1, group)
        g = order_fn(g)
        return g.reshape(-1, 64).max(axis=1).sum()

    base = cost(lambda g: g, 256)
    line = f"bounce {depth}: {n} rays, {s.mean():.1f} steps per ray; a wave pays {base / (n / 64):.1f} (all lanes busy: {s.sum() / 64 / (n / 64):.1f})"
    for group in (256, 4096):
        split = cost(lambda g: np.take_along_axis(g, np.argsort(g > 4, axis=1, kind="stable"), axis=1), group)
        full = cost(lambda g: np.sort(g, axis=1), group)
        line += f"; in {group}s: mesh / no mesh {split / base * 100:.0f} %, sorted by length {full / base * 100:.0f} %"
    print(line)
    for group in (256, 4096):  # the rays of a block / tile regrouped by the octant of their direction (order kept inside an octant)
        g = s.reshape(-1, group)
        o = octant.reshape(-1, group)
        by_octant = np.take_along_axis(g, np.argsort(o, axis=1, kind="stable"), axis=1).reshape(-1, 64).max(axis=1).sum()
        print(f"  regrouped by direction octant inside {group}s: {by_octant / base * 100:.0f} % of the wave-steps")
        key = o + 8 * (g > 4)  # ... and by "walks the mesh" as well (16 kinds)
        both = np.take_along_axis(g, np.argsort(key, axis=1, kind="stable"), axis=1).reshape(-1, 64).max(axis=1).sum()
        print(f"  by octant and mesh / no mesh inside {group}s: {both / base * 100:.0f} %")
    if depth == 0:
        # primary rays: a wave is 64 neighbours of a pixel row; what if it were a block of pixels?  (image cropped to 64s)
        k = n // (W * H)
        Wc, Hc = W - W % 64, H - H % 64
        img = s[:k * W * H].reshape(k, H, W)[:, :Hc, :Wc]
        rows = img.reshape(-1, 64).max(axis=1).sum()
        for bh, bw in ((8, 8), (4, 16), (16, 4), (2, 32)):
            blk = img.reshape(k, Hc // bh, bh, Wc // bw, bw).transpose(0, 1, 3, 2, 4).reshape(-1, 64).max(axis=1).sum()
            print(f"  primary rays in {bh} x {bw} pixel blocks per wave: {blk / rows * 100:.0f} % of the wave-steps of 1 x 64 strips")
        for bh, bw, ns in ((4, 4, 4), (2, 2, 16), (1, 1, 64), (4, 8, 2)):  # ... and several samples of a pixel in one wave
            kk = k - k % ns
            if kk == 0:
                continue
            sub = img[:kk]
            blk = (sub.reshape(kk // ns, ns, Hc // bh, bh, Wc // bw, bw).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 64).max(axis=1).sum())
            print(f"  {bh} x {bw} pixels x {ns} samples per wave: {blk / sub.reshape(-1, 64).max(axis=1).sum() * 100:.0f} %")
