"""Lane / wave step counts of the four-wide walks (needs a GPU and a library built with -DRT_EXP=32:
scripts/build_variant.sh prof -DRT_EXP=32; RT_MI355X_LIB=tmp_variants/lib_prof.so python scripts/exp_steps.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_knight
import raytracer_amd as rt
gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
lib = ctypes.CDLL(os.environ["RT_MI355X_LIB"])
lib.rt_exp_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
names = ["mesh node", "triangle", "world leaf", "world node"]
for scene in sys.argv[1:] or ["knight", "blocks_ball"]:
    sc = rt.Scene.load(os.path.join(ROOT, "examples", scene + ".json5"), ROOT)
    r = rt.Renderer(device=0, precision=0)
    r.upload(sc)
    W = 600
    H = sc.image_height(W)
    out = (ctypes.c_ulonglong * 8)()
    lib.rt_exp_prof(out, 1)
    _, _, stats = r.render_rows(W, H, 0, H, 10, 50, seed=1)
    lib.rt_exp_prof(out, 1)
    print(scene, "rays", stats.rays)
    for k, n in enumerate(names):
        lane, wave = out[2 * k], out[2 * k + 1]
        if wave:
            print(f"  {n:10s}: lane steps {lane:12d} ({lane / stats.rays:7.2f} / ray), wave steps {wave:11d}, lanes active {lane / (64.0 * wave):.3f}")
