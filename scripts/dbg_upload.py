"""Prints the traversal layout the library picks for each example scene (needs a GPU; RT_DEBUG=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ["RT_DEBUG"] = "1"
import gen_knight
import raytracer_amd as rt
gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
for name in ("two_spheres", "cornell", "globe", "blue_ball", "blocks_ball", "knight"):
    sc = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
    r = rt.Renderer(device=0, precision=0)
    sys.stderr.write(name + ": ")
    r.upload(sc)
