"""GPU tree build against the host's SAH regrouping (needs a GPU): upload times, node counts and the time of an image with
either tree, for the knight stand-in at several sizes.  usage: python scripts/exp_gpu_build.py [triangles ...]"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import gen_knight

sizes = [int(a) for a in sys.argv[1:]] or [100000, 1000000]
for tris in sizes:
    root = tempfile.mkdtemp(prefix="knight_")
    os.makedirs(os.path.join(root, "examples"))
    gen_knight.ensure(os.path.join(root, "examples", "knight.obj"), tris)
    shutil.copy(os.path.join(ROOT, "examples", "knight.json5"), os.path.join(root, "examples", "knight.json5"))
    results = {}
    for mode in ("0", "1"):
        os.environ["RT_GPU_BUILD"] = mode
        import raytracer_amd as rt
        sc = rt.Scene.load(os.path.join(root, "examples", "knight.json5"), root)
        r = rt.Renderer(device=0, precision=0, max_paths=1 << 26)
        t0 = time.time()
        r.upload(sc)
        t_up = (time.time() - t0) * 1e3
        up = r.upload_stats()
        W = 600
        H = sc.image_height(W)
        r.render_rows(W, H, 0, H, 16, 50, seed=1)  # warm-up
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            rgb8, sums, st = r.render_rows(W, H, 0, H, 100, 50, seed=1)
            best = min(best, st.ms_total)
        results[mode] = (rgb8, sums, st.rays)
        print(f"{tris} triangles, RT_GPU_BUILD={mode}: upload {t_up:.1f} ms (layout build {up.ms_build:.1f}, of it GPU trees {up.ms_gpu_build:.1f}; "
              f"trees built on the GPU {up.gpu_built_trees}, four-wide nodes {up.n_nodes4}), image {best:.2f} ms, rays {st.rays}", flush=True)
        r.close()
    same = np.array_equal(results["0"][1], results["1"][1]) and results["0"][2] == results["1"][2]
    print(f"  images and ray counts identical: {same}", flush=True)
    shutil.rmtree(root)
