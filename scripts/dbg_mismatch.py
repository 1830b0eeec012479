import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from oracle import binding as orc
name = sys.argv[1]
g = np.load(os.path.join(ROOT, "tests/golden", name + "_rays.npz"))
sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
r = rt.Renderer(0); r.upload(sc)
got = r.trace(g["rays"], int(g["seed"]))
want = g["hits"]
bad = np.nonzero((got["hit"] != want["hit"]) | (got["owner"] != want["owner"]) | (got["sub"] != want["sub"]))[0]
np.set_printoptions(precision=17, linewidth=200)
for i in bad[:10]:
    print("ray", i, g["rays"][i])
    print("  gpu ", got[i])
    print("  orc ", want[i])
