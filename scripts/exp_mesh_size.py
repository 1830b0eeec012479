"""Does the mesh walk's cost per step depend on whether the mesh fits the L2?  Renders the knight scene with
stand-in meshes of several sizes and prints extend time per ray and (with a -DRT_EXP=32 library) per node step.
usage (GPU): [RT_MI355X_LIB=tmp_variants/lib_prof.so] python scripts/exp_mesh_size.py"""
import ctypes, os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_knight
import raytracer_amd as rt

prof = None
if "RT_MI355X_LIB" in os.environ:
    lib = ctypes.CDLL(os.environ["RT_MI355X_LIB"])
    if hasattr(lib, "rt_exp_prof"):
        lib.rt_exp_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        prof = lib
for n in (6_000, 25_000, 100_000, 400_000, 1_600_000):
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "examples"))
    v, nn, f = gen_knight.build(n)
    gen_knight.write_obj(os.path.join(d, "examples", "knight.obj"), v, nn, f)
    shutil.copy(os.path.join(ROOT, "examples", "knight.json5"), os.path.join(d, "examples", "knight.json5"))
    sc = rt.Scene.load(os.path.join(d, "examples", "knight.json5"), d)
    r = rt.Renderer(device=0, precision=0)
    r.upload(sc)
    W = 600
    H = sc.image_height(W)
    r.render_rows(W, H, 0, H, 4, 50, seed=1)  # warm-up
    out = (ctypes.c_ulonglong * 8)()
    if prof:
        prof.rt_exp_prof(out, 1)
    _, _, st = r.render_rows(W, H, 0, H, 20, 50, seed=1)
    line = f"{len(f):8d} triangles: {st.rays / 1e6:6.1f} Mrays, extend {st.ms_extend:7.2f} ms = {st.ms_extend * 1e6 / st.rays:6.2f} ns/ray"
    if prof:
        prof.rt_exp_prof(out, 1)
        line += f", {out[0] / st.rays:5.2f} mesh node steps / ray ({out[0] / (64.0 * out[1]):.2f} lanes), {out[2] / st.rays:5.2f} triangle tests / ray ({out[2] / (64.0 * out[3]):.3f})"
    print(line, flush=True)
    shutil.rmtree(d)
