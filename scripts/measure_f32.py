"""The fast mode (RT_PRECISION_F32: the same kernels in f32, boxes rounded outwards) against the f64 oracle, measured (SURVEY.md
section 7, "Hard parts": "a fast mode (all-f32) is measured separately and its ID-mismatch rate reported").  Run on the GPU box.

Per scene, on the committed golden rays (tests/golden/<scene>_rays.npz: primary and secondary rays with the oracle's records):
hit / miss flips, the primitive-ID mismatch rate among rays that are not exact ties, the largest relative error of t where the
IDs agree; and on a 160-px frame of primary rays the same.  Prints one JSON object (profiles/<round>/f32_mode.json)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_knight  # noqa: E402
import raytracer_amd as rt  # noqa: E402

gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
TIE_REL = 1e-9


def compare(got, want, second_t):
    hit = want["hit"] == 1
    flips = int((got["hit"] != want["hit"]).sum())
    both = hit & (got["hit"] == 1)
    tie = np.abs(second_t - want["t"]) <= TIE_REL * np.maximum(np.abs(want["t"]), 1e-300)
    same = (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    judged = both & ~tie
    wrong = judged & ~same
    ok = both & same
    t_err = np.abs(got["t"][ok] - want["t"][ok]) / np.maximum(np.abs(want["t"][ok]), 1e-300)
    # a wrong ID that is a near-tie in f32 terms: the two candidates are within 1e-5 relative of each other
    near = wrong & (np.abs(second_t - want["t"]) <= 1e-5 * np.maximum(np.abs(want["t"]), 1e-300))
    return {"rays": int(len(want)), "hits": int(hit.sum()), "hit_miss_flips": flips, "judged": int(judged.sum()), "id_mismatches": int(wrong.sum()),
            "id_mismatch_rate": float(wrong.sum() / max(1, judged.sum())), "of_which_within_1e-5_of_the_second_candidate": int(near.sum()),
            "max_rel_t_error": float(t_err.max(initial=0)), "p99_rel_t_error": float(np.quantile(t_err, 0.99)) if len(t_err) else 0.0}


out = {}
for name in ["two_spheres", "cornell", "globe", "blocks_ball", "knight", "blue_ball"]:
    path = os.path.join(ROOT, "examples", name + ".json5")
    sc = rt.Scene.load(path, ROOT)
    r32 = rt.Renderer(0, precision=rt.PRECISION_F32)
    r32.upload(sc)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + "_rays.npz"))
    res = {"golden_rays": compare(r32.trace(g["rays"], seed=int(g["seed"])), g["hits"], g["hits"]["second_t"])}
    # a frame of primary rays: the f64 library stands in for the oracle here (it equals it ray for ray: tests/test_gpu_parity.py)
    r64 = rt.Renderer(0)
    r64.upload(sc)
    W = 160
    H = sc.image_height(W)
    rays = r64.primary_rays(W, H, 0, H, 0, 1)
    want = r64.trace(rays, 1)
    res["primary_160px"] = compare(r32.trace(rays, 1), want, want["second_t"])
    out[name] = res
    r32.close()
    r64.close()
print(json.dumps(out, indent=1))
