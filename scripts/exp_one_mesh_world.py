import os, re, sys
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
text = open("examples/knight.json5").read()
text, n = re.subn(r"world:\s*\[[^\]]*\]", 'world: ["board", "knight"]', text)
sc = rt.Scene.from_text(text, os.getcwd())
for env in ({}, {"RT_BVH4": "0"}):
    os.environ.pop("RT_BVH4", None)
    os.environ.update(env)
    r = rt.Renderer(0, max_paths=1 << 26)
    r.upload(sc)
    W = 600; H = sc.image_height(W)
    r.render_rows(W, H, 0, H, 16, 50, seed=1)
    best = min(r.render_rows(W, H, 0, H, 100, 50, seed=1)[2].ms_total for _ in range(3))
    print(env, "nodes4", r.upload_stats().n_nodes4, "image %.2f ms" % best, flush=True)
    r.close()
