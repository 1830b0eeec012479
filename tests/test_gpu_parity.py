"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same inputs and against the committed golden fixtures.

Bars (BASELINE.json north_star):
  * hit identity (owner, sub, world entry, face, material): bit-exact for every ray that is not an
    exact tie (a tie = the oracle's nearest other candidate within 1e-9 relative of the hit; on a
    tie the GPU must return one of the tied primitives - in practice it returns the same one);
  * hit distance / normal / point / uv: <= 1e-5 relative (measured: ~1e-16, the f64 kernels follow
    the reference's operation order);
  * images: the GPU and the oracle walk the same random streams, so per-pixel radiance sums agree
    to rounding except where a last-bit difference (libm vs OCML sin/acos/pow/log) flips a branch;
    required: >= 99% of pixels within rtol 1e-6, and the image means within 1%.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SCENES, assert_image_parity

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5       # north_star: hit t / normal within 1e-5 relative
TIE_REL = 1e-9       # tie rule


def tie_mask(h, second_t=None):
    second_t = h["second_t"] if second_t is None else second_t
    return (h["hit"] == 1) & (np.abs(second_t - h["t"]) <= TIE_REL * np.maximum(np.abs(h["t"]), 1e-300))


def check_hits(got, want, second_t=None):
    """got: rt_trace's records; want: the oracle's (second_t: its nearest other candidate, None for the far-origin test).
    The tie mask comes from the LIBRARY's own rt_hit.second_t and must be the oracle's."""
    assert np.array_equal(got["hit"], want["hit"]), "hit / miss differs"
    hit = want["hit"] == 1
    same = (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    tie = tie_mask(got)
    assert np.all(np.isposinf(got["second_t"][~hit])) and np.all(got["inst_id"][~hit] == -1)
    assert np.all(got["inst_id"][hit] >= 0)
    assert np.all(got["second_t"][hit] >= got["t"][hit] * (1 - TIE_REL)), "a candidate nearer than the hit"
    if second_t is not None:
        assert np.array_equal(tie, tie_mask(want, second_t)), "the library's tie flags are not the oracle's"
        # the same second candidate where the two agree on the first (an exact tie names the same distance either way)
        both = hit & same
        finite = both & np.isfinite(second_t)
        assert np.array_equal(np.isfinite(got["second_t"][both]), np.isfinite(second_t[both]))
        err = np.abs(got["second_t"][finite] - second_t[finite]) / np.maximum(np.abs(second_t[finite]), 1e-300)
        assert err.max(initial=0) <= REL_TOL, "second_t"
    assert np.all(same[hit & ~tie]), f"{(~same[hit & ~tie]).sum()} primitive ids differ on non-tied rays"
    # on a tie the GPU must still return one of the tied primitives: same distance
    tied = hit & tie & ~same
    assert np.all(np.abs(got["t"][tied] - want["t"][tied]) <= TIE_REL * np.maximum(np.abs(want["t"][tied]), 1e-300))
    ok = hit & same
    for f in ["world_index", "front_face", "material"]:
        assert np.array_equal(got[f][ok], want[f][ok]), f
    t_err = np.abs(got["t"][ok] - want["t"][ok]) / np.maximum(np.abs(want["t"][ok]), 1e-300)
    assert t_err.max(initial=0) <= REL_TOL
    assert np.abs(got["n"][ok] - want["n"][ok]).max(initial=0) <= REL_TOL
    scale = np.maximum(np.abs(want["p"][ok]).max(axis=1, initial=1.0), 1.0)
    assert (np.abs(got["p"][ok] - want["p"][ok]).max(axis=1) / scale).max(initial=0) <= REL_TOL
    assert np.abs(got["u"][ok] - want["u"][ok]).max(initial=0) <= REL_TOL
    assert np.abs(got["v"][ok] - want["v"][ok]).max(initial=0) <= REL_TOL
    return int(tie.sum())


@pytest.mark.parametrize("name", SCENES)
def test_trace_matches_golden_fixture(gpu_scenes, name):
    g = np.load(os.path.join(GOLDEN, name + "_rays.npz"))
    _, r = gpu_scenes(name)
    got = r.trace(g["rays"], seed=int(g["seed"]))
    check_hits(got, g["hits"], g["hits"]["second_t"])


@pytest.mark.parametrize("name", SCENES)
def test_raygen_and_primary_hits_match_oracle(gpu_scenes, oracle_scenes, name):
    sc, r = gpu_scenes(name)
    o = oracle_scenes(name)
    W = 160
    H = sc.image_height(W)
    assert H == o.image_height(W)
    for sample, seed in [(0, 1), (5, 99)]:
        rays = r.primary_rays(W, H, 0, H, sample, seed)
        want = o.primary_rays(W, H, 0, H, sample, seed)
        for f in ["o", "d", "time", "tmin", "tmax"]:
            assert np.allclose(rays[f], want[f], rtol=1e-14, atol=1e-14), f
        hits = o.trace(want, seed)
        check_hits(r.trace(want, seed), hits, hits["second_t"])


@pytest.mark.parametrize("name", SCENES)
def test_secondary_rays_match_oracle(gpu_scenes, oracle_scenes, name):
    # rays leaving surfaces in random directions, at random times, incl. grazing / inside-medium cases
    sc, r = gpu_scenes(name)
    o = oracle_scenes(name)
    rng = np.random.default_rng(42)
    W = 64
    H = sc.image_height(W)
    prim = o.primary_rays(W, H, 0, H, 1, 3)
    ph = o.trace(prim, 3)
    ok = ph["hit"] == 1
    n = int(ok.sum())
    rays = np.zeros(3 * n, dtype=prim.dtype)
    rays["o"] = np.tile(ph["p"][ok], (3, 1))
    d = rng.normal(size=(3 * n, 3))
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["time"] = rng.uniform(0, 1, 3 * n)
    rays["tmin"] = 0.001
    rays["tmax"] = np.inf
    rays["tmax"][: n // 2] = rng.uniform(0.1, 500.0, n // 2)   # finite upper bounds
    want = o.trace(rays, 77)
    check_hits(r.trace(rays, 77), want, want["second_t"])


def test_trace_edge_cases(gpu_scenes, oracle_scenes, rt):
    sc, r = gpu_scenes("cornell")
    o = oracle_scenes("cornell")
    assert len(r.trace(np.zeros(0, dtype=rt.RAY_DTYPE))) == 0              # empty input
    rays = o.primary_rays(64, 64, 0, 64, 0, 5)
    for n in [1, 63, 64, 65, 257]:                                         # ragged wave / block tails
        want = o.trace(rays[:n], 5)
        check_hits(r.trace(rays[:n], 5), want, want["second_t"])
    # un-normalised directions are normalised as Ray::new does; misses stay misses
    odd = rays[:100].copy()
    odd["d"] *= 7.5
    odd["o"][:50] = [278, 278, -800]
    odd["d"][:50] = [0, 0, -1]                                             # looking away from the box
    want = o.trace(odd, 5)
    got = r.trace(odd, 5)
    check_hits(got, want, want["second_t"])
    assert got["hit"][:50].sum() == 0


def test_exact_ties_follow_the_reference_rules(rt, orc):
    # Two coincident rects listed in `world`: the BVH's later leaf wins an exact tie (a15); the two
    # coincident sides of touching blocks inside one Block list: the first wins (a12).
    text = """{
      camera: {lookFrom: [0,0,-5], lookAt: [0,0,0], directionUp: [0,1,0], fieldOfView: 40, aspectRatio: [1,1],
               aperture: 0, distanceToFocus: 5, startTime: 0, endTime: 1},
      background: {type: "plainColour", colour: [0,0,0]},
      textures: {t: {type: "solidColour", colour: [1,1,1]}},
      materials: {m: {type: "lambertian", texture: "t"}, k: {type: "metal", albedo: [1,1,1], fuzz: 0}},
      objects: {
        a: {type: "rect", corner0: [-1,-1,1], corner1: [1,1,1], facingForward: false, material: "m"},
        b: {type: "rect", corner0: [-1,-1,1], corner1: [1,1,1], facingForward: false, material: "k"},
        c: {type: "block", corner0: [-1,-1,3], corner1: [1,1,4], material: "m"},
        d: {type: "block", corner0: [-1,-1,4], corner1: [1,1,5], material: "k"},
      },
      world: ["a", "b", "c", "d"],
    }"""
    scene = rt.Scene.from_text(text)
    o = orc.OracleScene.from_text(text)
    r = rt.Renderer(0)
    r.upload(scene)
    rays = np.zeros(3, dtype=rt.RAY_DTYPE)
    rays["o"] = [[0.1, 0.2, -5], [0.1, 0.2, 3.5], [0.3, -0.4, 1.5]]
    rays["d"] = [[0, 0, 1], [0, 0, 1], [0, 0, 1]]
    rays["tmin"] = 0.001
    rays["tmax"] = np.inf
    want = o.trace(rays, 0)
    got = r.trace(rays, 0)
    assert np.all(want["second_t"][:2] == want["t"][:2])      # these are exact ties in the oracle
    for f in ["hit", "owner", "sub", "material", "front_face"]:
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["t"], want["t"])
    # the library says so itself (rt_hit.second_t): exact ties on the first two rays, not on the third
    assert np.array_equal(got["second_t"], want["second_t"])
    assert np.all(got["second_t"][:2] == got["t"][:2]) and got["second_t"][2] > got["t"][2]
    assert check_hits(got, want, want["second_t"]) == 2


@pytest.mark.parametrize("name", SCENES)
def test_inst_id_names_the_instance_that_was_hit(gpu_scenes, name):
    # rt_hit.inst_id indexes rt_scene_desc.objects: that instance lists the primitive (owner, sub) and came through world_index
    from test_scene_loader import _desc

    g = np.load(os.path.join(GOLDEN, name + "_rays.npz"))
    sc, r = gpu_scenes(name)
    d = _desc(sc)
    got = r.trace(g["rays"], seed=int(g["seed"]))
    objs = d["objects"]
    for h in got[got["hit"] == 1][::7]:
        o = objs[int(h["inst_id"])]
        assert o.world_index == h["world_index"]
        if o.kind == 2:   # RT_OBJ_RECTS: one of the instance's rects
            assert any(d["rects"][o.first + k].owner == h["owner"] and d["rects"][o.first + k].sub == h["sub"] for k in range(o.count))
        elif o.kind == 0:
            assert d["spheres"][o.first].owner == h["owner"]


@pytest.mark.parametrize("name", SCENES)
def test_render_matches_golden_image(gpu_scenes, name):
    g = np.load(os.path.join(GOLDEN, name + "_image.npz"))
    W, H = int(g["width"]), int(g["height"])
    _, r = gpu_scenes(name)
    rgb8, sums, stats = r.render_rows(W, H, 0, H, int(g["spp"]), int(g["max_bounces"]), seed=int(g["seed"]))
    assert stats.paths == W * H * int(g["spp"])
    agree = assert_image_parity(sums, g["sums"], stats.rays, int(g["rays"]), int(g["spp"]), int(g["max_bounces"]), label=name)
    assert (rgb8 != g["rgb8"]).mean() <= 1.0 - agree   # bytes can only differ where sums do


@pytest.mark.parametrize("name", ["cornell", "two_spheres", "blocks_ball"])
def test_render_matches_live_oracle(gpu_scenes, oracle_scenes, name):
    sc, r = gpu_scenes(name)
    o = oracle_scenes(name)
    W = 72
    H = sc.image_height(W)
    rgb8, sums, stats = r.render_rows(W, H, 0, H, 16, 50, seed=2024)
    o8, osums, orays = o.render(W, H, 0, H, 16, 50, seed=2024, threads=8)
    agree = assert_image_parity(sums, osums, stats.rays, orays, 16, 50, label=name,
                                noise=lambda: o.render(W, H, 0, H, 16, 50, seed=2025, threads=8)[1])
    assert (rgb8 != o8).mean() <= 1.0 - agree


def test_render_properties_at_full_size(gpu_scenes):
    # BASELINE configs[1] at full size (600 x 600, 100 spp, 50 bounces): properties that do not need
    # the oracle.  (i) sample ranges add up: 100 spp == 37 spp + 63 spp of the same streams;
    # (ii) row bands are independent: a band rendered alone equals the same rows of the full render;
    # (iii) the result is finite, non-negative and reproducible run to run (bit-exact).
    import torch

    sc, r = gpu_scenes("cornell")
    W, H, spp, mb = 600, 600, 100, 50
    dev = torch.device("cuda", 0)
    full = torch.zeros(W * H * 3, dtype=torch.float64, device=dev)
    parts = torch.zeros_like(full)
    again = torch.zeros_like(full)
    torch.cuda.synchronize()
    st = r.render_accum_device(W, H, 0, H, 0, spp, mb, 7, full.data_ptr())
    r.render_accum_device(W, H, 0, H, 0, 37, mb, 7, parts.data_ptr())
    r.render_accum_device(W, H, 0, H, 37, 63, mb, 7, parts.data_ptr())
    r.render_accum_device(W, H, 0, H, 0, spp, mb, 7, again.data_ptr())
    assert st.paths == W * H * spp and st.rays > 5 * st.paths
    assert torch.equal(full, again)
    assert torch.isfinite(full).all() and (full >= 0).all()
    assert torch.allclose(full, parts, rtol=1e-12, atol=1e-12)
    band = torch.zeros(W * 40 * 3, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    r.render_accum_device(W, H, 300, 340, 0, spp, mb, 7, band.data_ptr())
    assert torch.equal(band, full.view(H, W * 3)[300:340].reshape(-1))
    # quantisation on device equals the reference formula applied on the host
    rgb8 = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    r.quantize_device(full.data_ptr(), W * H, spp, rgb8.data_ptr())
    host = np.floor(255.0 * np.clip(np.abs(np.sqrt(full.cpu().numpy() / spp)), 0, 0.999)).astype(np.uint8)
    assert np.array_equal(rgb8.cpu().numpy(), host)
    assert 40 < host.mean() < 200      # a lit cornell box, not black / not blown out


def test_batches_do_not_change_the_image(rt, gpu_scenes):
    # a small path budget forces several batches and several row bands; the image must not change
    sc, ref = gpu_scenes("two_spheres")
    small = rt.Renderer(0, max_paths=5000)
    small.upload(sc)
    W = 96
    H = sc.image_height(W)
    a8, a, sa = ref.render_rows(W, H, 0, H, 9, 20, seed=5)
    b8, b, sb = small.render_rows(W, H, 0, H, 9, 20, seed=5)
    assert sb.batches > sa.batches
    assert sa.rays == sb.rays
    assert np.allclose(a, b, rtol=1e-13, atol=0) and np.array_equal(a8, b8)


def test_f32_fast_mode_is_statistically_equivalent(rt, gpu_scenes):
    # The declared fast mode (all-f32) is compared statistically only: same image up to Monte-Carlo noise.
    sc, r64 = gpu_scenes("cornell")
    _, r32 = gpu_scenes("cornell", precision=rt.PRECISION_F32)
    W = 64
    _, a, _ = r64.render_rows(W, W, 0, W, 256, 50, seed=1)
    _, b, _ = r32.render_rows(W, W, 0, W, 256, 50, seed=2)
    _, c, _ = r64.render_rows(W, W, 0, W, 256, 50, seed=3)
    noise = np.abs(a - c).mean()
    assert np.abs(a - b).mean() <= 1.25 * noise          # within 1.25x of oracle-vs-oracle error
    assert abs(a.mean() - b.mean()) <= 0.03 * a.mean()


def test_bad_arguments_are_errors_not_faults(rt, gpu_scenes):
    sc, r = gpu_scenes("cornell")
    fresh = rt.Renderer(0)
    with pytest.raises(rt.RtError) as e:
        fresh.render_rows(8, 8, 0, 8, 1, 1)
    assert e.value.code == -3            # RT_ERR_NO_SCENE
    with pytest.raises(rt.RtError):
        r.render_rows(1, 8, 0, 8, 1, 1)  # width - 1 == 0 would divide by zero (src/main.rs:132)
    with pytest.raises(rt.RtError):
        r.render_rows(8, 8, 5, 3, 1, 1)
    with pytest.raises(rt.RtError):
        r.render_rows(8, 8, 0, 9, 1, 1)
    rgb8, sums, st = r.render_rows(8, 8, 0, 8, 2, 0)   # max_bounces 0: cast_ray returns black at once
    assert st.rays == 0 and not sums.any() and not rgb8.any()


# ---------------------------------------------------------------- every traversal layout gives the same answer
# The library picks a traversal per scene (sweep of a few objects, own-box nodes in LDS, DNode2 tree
# walk nearer-child-first, fused extend+shade).  They are different programs for the same function:
# the images must agree bit for bit, and so must the hits.  The environment switches are read ONCE, when a
# context is created (capi.cpp, struct Tuning): every case makes its own.
@pytest.mark.parametrize(
    "name, env",
    [
        ("cornell", {"RT_SWEEP_MAX": "0"}),                    # own-box nodes in LDS instead of the sweep
        ("cornell", {"RT_FUSE": "0"}),                         # extend, then shade, instead of k_bounce
        ("cornell", {"RT_SWEEP_MAX": "0", "RT_FUSE": "1"}),    # fused kernel over the LDS tree walk
        ("blue_ball", {"RT_SWEEP_MAX": "0"}),
        ("blocks_ball", {"RT_ORDERED": "0"}),                  # reference (left-first) order in the DNode2 walk
        ("blocks_ball", {"RT_FUSE": "1"}),
        ("blocks_ball", {"RT_BVH4": "0"}),                     # binary instead of four-wide world tree
        ("knight", {"RT_ORDERED": "0"}),
        ("knight", {"RT_BVH4": "0"}),                          # binary instead of four-wide mesh tree
        ("knight", {"RT_BVH4": "0", "RT_ORDERED": "0"}),
        ("knight", {"RT_SAH": "0"}),                           # the reference's grouping (folded) instead of SAH trees
        ("blocks_ball", {"RT_SAH": "0"}),
        ("knight", {"RT_SAH": "0", "RT_ORDERED": "0"}),
        ("knight", {"RT_REFILL": "16"}),                        # lanes take new rays while their wave still walks
        ("knight", {"RT_REFILL": "64", "RT_FEATURES": "all"}),
        ("blocks_ball", {"RT_REFILL": "64", "RT_FUSE": "0"}),   # two kernels per bounce, k_extend_refill instead of k_extend
        ("blocks_ball", {"RT_REFILL": "5", "RT_STACK_CAP": "3", "RT_FUSE": "0"}),
        ("knight", {"RT_FEATURES": "all"}),                     # the general four-wide kernel instead of the one without media
        ("blocks_ball", {"RT_FEATURES": "all"}),                # ... instead of the one without meshes
        ("knight", {"RT_STACK_CAP": "6"}),                   # four-wide walk mostly on the private overflow stack
        ("blocks_ball", {"RT_STACK_CAP": "3"}),
        ("globe", {"RT_SWEEP_MAX": "0", "RT_FUSE": "0"}),
        ("cornell", {"RT_TAIL": "0"}),                         # a launch per bounce to the end, no tail launch
        ("knight", {"RT_TAIL": "0"}),
        ("blocks_ball", {"RT_TAIL": "0"}),
        ("blue_ball", {"RT_TAIL": "1000000000"}),              # the whole render after bounce 4 in the tail launch
        ("cornell", {"RT_PIXEL_BLOCKS": "2"}),                 # a sweep scene with the tree scenes' queue order (pixels x samples per wave)
        ("knight", {"RT_PIXEL_BLOCKS": "1"}),                  # 8 x 8 blocks of pixels, sample by sample
        ("blocks_ball", {"RT_PIXEL_BLOCKS": "1"}),
        ("cornell", {"RT_PIXEL_BLOCKS": "0"}),                 # paths enter the queue along the pixel rows
        ("knight", {"RT_PIXEL_BLOCKS": "0"}),
        ("blocks_ball", {"RT_PIXEL_BLOCKS": "0"}),             # 96 x 54: a ragged last strip of 6 rows
        ("blocks_ball", {"RT_COOP": "0"}),                     # world tree without meshes: the world walk is what the groups share
        ("blocks_ball", {"RT_COOP": "32"}),
        ("blocks_ball", {"RT_COOP": "32", "RT_FEATURES": "all"}),
        ("knight", {"RT_COOP": "0"}),                          # one lane per path to the end of the tail launch (no lane groups)
        ("knight", {"RT_COOP": "32"}),                         # groups of two lanes and up
        ("knight", {"RT_COOP": "32", "RT_FEATURES": "all"}),
        ("knight", {"RT_COOP": "32", "RT_STACK_CAP": "20"}),   # a frontier with little room: mostly one entry a round
        ("cornell", {"RT_TAIL": "0", "RT_FUSE_RAYGEN": "0"}),   # (RT_TAIL=0 above: the first bounce makes its camera rays itself; here k_raygen does)
        ("globe", {"RT_TAIL": "0"}),
        ("cornell", {"RT_TAIL": "0", "RT_PIXEL_BLOCKS": "2"}),  # ... in the pixels-by-samples order of the queue
        ("blue_ball", {"RT_TAIL": "0", "RT_FUSE": "1"}),        # a sweep that is not lean, fused: k_bounce<16, WALK_SWEEP, .., FIRST>
        ("knight", {"RT_GPU_BUILD": "1"}),                     # the mesh's four-wide tree from the GPU builder (Morton order, radix tree, fold)
        ("knight", {"RT_GPU_BUILD": "1", "RT_COOP": "32"}),
        ("knight", {"RT_GPU_BUILD": "1", "RT_ORDERED": "0", "RT_FEATURES": "all"}),
        ("knight", {"RT_LEAF_VOTE": "1"}),                     # every mesh leaf tested at once (no waiting for other lanes' leaves)
        ("knight", {"RT_LEAF_FRAC": "16"}),                    # ... only when every lane still walking holds one
        ("knight", {"RT_LEAF_FRAC": "0", "RT_LEAF_VOTE": "5", "RT_FEATURES": "all"}),
        ("two_spheres", {"RT_FUSE": "0"}),                     # k_extend + k_shade instead of the fused bounce of a sweep without meshes or media (WALK_SWEEP_FLAT)
        ("two_spheres", {"RT_TAIL": "0", "RT_FUSE_RAYGEN": "0"}),
        ("two_spheres", {"RT_TAIL": "100000000"}),             # every bounce in the tail launch
        ("blue_ball", {"RT_FUSE": "0"}),                       # the general sweep, two kernels per bounce
        ("cornell", {"RT_COOP": "0"}),                         # no lane groups at the end of the tail launch (a swept world of rect lists shares a path's rects out)
        ("cornell", {"RT_COOP": "32", "RT_TAIL": "100000000"}),  # ... groups from 32 live paths down, every bounce in the tail launch
        ("knight", {"RT_OBJ_FRAC": "0"}),                      # the world tree's leaves tested at once / when every walking lane holds one
        ("blocks_ball", {"RT_OBJ_FRAC": "16", "RT_FUSE": "0"}),   # k_extend + k_shade (world trees without meshes are fused by default)
        ("blocks_ball", {"RT_OBJ_FRAC": "0", "RT_REFILL": "0", "RT_COOP": "0"}),
    ],
)
def test_traversal_layouts_agree_bit_for_bit(rt, gpu_scenes, name, env, monkeypatch):
    sc, default_renderer = gpu_scenes(name)
    W = 96
    H = sc.image_height(W)
    rgb8_a, sums_a, st_a = default_renderer.render_rows(W, H, 0, H, 8, 20, seed=77)
    rays = np.zeros(4096, dtype=rt.RAY_DTYPE)
    prim = default_renderer.primary_rays(W, H, 0, H, 0, seed=77)
    rays[:] = prim[np.linspace(0, len(prim) - 1, 4096).astype(int)]
    hits_a = default_renderer.trace(rays, seed=5)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = rt.Renderer(device=0, precision=0)
    other.upload(sc)
    if "RT_GPU_BUILD" in env:
        assert other.upload_stats().gpu_built_trees == 1 and default_renderer.upload_stats().gpu_built_trees == 0
    rgb8_b, sums_b, st_b = other.render_rows(W, H, 0, H, 8, 20, seed=77)
    hits_b = other.trace(rays, seed=5)
    assert st_a.rays == st_b.rays
    assert st_a.max_depth_reached == st_b.max_depth_reached
    assert np.array_equal(sums_a, sums_b)
    assert np.array_equal(rgb8_a, rgb8_b)
    for field in ("hit", "owner", "sub", "world_index", "front_face", "material", "t", "p", "n", "u", "v"):
        assert np.array_equal(hits_a[field], hits_b[field]), field


@pytest.mark.parametrize("name,width,rows,spp", [("knight", 104, (13, 50), 67), ("blocks_ball", 96, (0, 54), 5), ("knight", 64, (8, 24), 130),
                                                 ("blocks_ball", 100, (3, 40), 9)])
def test_the_order_of_the_queue_does_not_change_the_image(rt, gpu_scenes, name, width, rows, spp, monkeypatch):
    # k_raygen puts the paths into the queue in one of three orders (DBatch::blocked): along the pixel rows, through
    # 8 x 8 blocks of pixels, or n samples each of 64 / n neighbouring pixels per wave (runs of 64, 32, ... samples;
    # rows that do not fill a strip of 8 at the end; a width that is not a multiple of 8 falls back to rows).  A path
    # is named by its row-major id in all of them, so sums, bytes and ray counts must be the same to the bit.
    sc, _ = gpu_scenes(name)
    H = sc.image_height(width)
    j0, j1 = rows[0], min(rows[1], H)
    results = []
    for order in ("0", "1", None):
        if order is None:
            monkeypatch.delenv("RT_PIXEL_BLOCKS", raising=False)
        else:
            monkeypatch.setenv("RT_PIXEL_BLOCKS", order)
        r = rt.Renderer(device=0, precision=0)   # (the switches are read when a context is created)
        r.upload(sc)
        results.append(r.render_rows(width, H, j0, j1, spp, 12, seed=31))
        r.close()
    for rgb8, sums, st in results[1:]:
        assert st.rays == results[0][2].rays
        assert np.array_equal(sums, results[0][1])
        assert np.array_equal(rgb8, results[0][0])


def test_checkers_that_emit_and_checkers_that_do_not(rt, orc):
    # Checkered::emitted picks a part with sin(k u) sin(k v) (no PI), scatter with sin(k PI u) sin(k PI v)
    # (src/materials.rs:166-183).  The kernel skips the emitted() selection for a checker none of whose
    # parts - through nested checkers - is a light: both kinds, nested, against the oracle.
    text = """{
      camera: {lookFrom: [0,1.5,-6], lookAt: [0,0.5,0], directionUp: [0,1,0], fieldOfView: 45, aspectRatio: [1,1],
               aperture: 0, distanceToFocus: 6, startTime: 0, endTime: 1},
      background: {type: "plainColour", colour: [0.05,0.05,0.1]},
      textures: {w: {type: "solidColour", colour: [0.8,0.8,0.8]}, g: {type: "solidColour", colour: [0.2,0.6,0.3]},
                 l: {type: "solidColour", colour: [3,2,1]}},
      materials: {white: {type: "lambertian", texture: "w"}, green: {type: "lambertian", texture: "g"},
                  lamp: {type: "diffuseLight", emit: "l"},
                  dull: {type: "checkered", odd: "white", even: "green", tileDensity: 9},
                  glow: {type: "checkered", odd: "lamp", even: "green", tileDensity: 7},
                  deep: {type: "checkered", odd: "dull", even: "glow", tileDensity: 3}},
      objects: {
        floor: {type: "rect", corner0: [-4,0,-4], corner1: [4,0,4], facingForward: true, material: "dull"},
        a: {type: "sphere", centre: [-1.2,1,0], radius: 1, material: "glow"},
        b: {type: "sphere", centre: [1.2,1,0], radius: 1, material: "deep"},
      },
      world: ["floor", "a", "b"],
    }"""
    scene = rt.Scene.from_text(text)
    o = orc.OracleScene.from_text(text)
    r = rt.Renderer(0)
    r.upload(scene)
    W = H = 64
    rgb8, sums, st = r.render_rows(W, H, 0, H, 16, 12, seed=9)
    o8, osums, orays = o.render(W, H, 0, H, 16, 12, seed=9, threads=4)
    assert_image_parity(sums, osums, st.rays, orays, 16, 12, label="checkers",
                        noise=lambda: o.render(W, H, 0, H, 16, 12, seed=10, threads=4)[1])
    assert sums.max() > 1.0  # the lamps are seen


def _corners_scene(tmp, with_mesh, extra_spheres, nested=False):
    """Every object kind and wrapper of the schema in one world: moving sphere, lone
    triangle, plane, rotateX / rotateY / rotateZ around translate, spotlight, media bounded by a sphere, a block
    and (with_mesh) a mesh, nested checker, image-free materials of every type."""
    objs = """
        ball: {type: "sphere", centre: [0,1,0], radius: 1, material: "glass"},
        mover: {type: "movingSphere", centre0: [-2.5,0.6,0.5], centre1: [-2.5,1.4,0.5], time0: 0, time1: 1, radius: 0.6, material: "red"},
        tri: {type: "triangle", point0: [1.5,0,1.5], point1: [3.5,0,1.0], point2: [2.5,2.2,1.8], material: "mirror"},
        crate: {type: "block", corner0: [-0.5,-0.5,-0.5], corner1: [0.5,0.5,0.5], material: "check"},
        crateX: {type: "rotateX", prototype: "crate", degrees: 25},
        crateXY: {type: "rotateY", prototype: "crateX", degrees: 40},
        crateXYZ: {type: "rotateZ", prototype: "crateXY", degrees: -15},
        crateAt: {type: "translate", prototype: "crateXYZ", offset: [2.2,0.8,-1.2]},
        fogBall: {type: "sphere", centre: [-1.5,1.2,-1.5], radius: 0.9, material: "glass"},
        fog: {type: "constantMedium", boundary: "fogBall", phaseFunction: "mist", density: 1.5},
        hazeBox: {type: "block", corner0: [-6,-1,-6], corner1: [6,6,6], material: "glass"},
        haze: {type: "constantMedium", boundary: "hazeBox", phaseFunction: "mist", density: 0.02},
        lamp: {type: "spotlight", lookFrom: [0,5,-1], lookAt: [0,0,0], length: 1.0, width: 1.5, light: [9,8,7]},
        ground: {type: "plane", point0: [0,0,0], point1: [1,0,0], point2: [0,0,1], uvRepeat: 2.0, material: "check"},
    """
    world = ["ball", "mover", "tri", "crateAt", "fog", "haze", "lamp", "ground"]
    if nested:
        # a medium whose boundary is a medium (src/objects.rs:404-409: any Hittable), the inner one also in the world, a
        # second pair behind a transform
        objs += """
        fogOfFog: {type: "constantMedium", boundary: "fog", phaseFunction: "smoke", density: 2.0},
        puffBall: {type: "sphere", centre: [0,0,0], radius: 0.8, material: "glass"},
        puff: {type: "constantMedium", boundary: "puffBall", phaseFunction: "mist", density: 0.9},
        puffAt: {type: "translate", prototype: "puff", offset: [2.6,2.6,0.4]},
        puffOfPuff: {type: "constantMedium", boundary: "puffAt", phaseFunction: "smoke", density: 3.0},
        puffTurned: {type: "rotateY", prototype: "puffOfPuff", degrees: 30},
        """
        world += ["fogOfFog", "puffTurned"]
    if with_mesh:
        obj = os.path.join(tmp, "tetra.obj")
        with open(obj, "w") as f:
            f.write("v 0 0 0\nv 1.6 0 0\nv 0.8 0 1.4\nv 0.8 1.5 0.5\nf 1 3 2\nf 1 2 4\nf 2 3 4\nf 3 1 4\n")
        objs += """
        tetra: {type: "mesh", filename: "%s", material: "red"},
        tetraAt: {type: "translate", prototype: "tetra", offset: [-0.5,0,-3.2]},
        cloudShape: {type: "translate", prototype: "tetra", offset: [0.8,0,2.2]},
        cloud: {type: "constantMedium", boundary: "cloudShape", phaseFunction: "mist", density: 2.5},
        """ % obj
        world += ["tetraAt", "cloud"]
    for k in range(extra_spheres):
        objs += 'p%d: {type: "sphere", centre: [%g,0.25,%g], radius: 0.25, material: "%s"},\n' % (
            k, -4.0 + 0.6 * (k % 15), 2.6 + 0.6 * (k // 15), ["red", "mirror", "glass"][k % 3])
        world.append("p%d" % k)
    return """{
      camera: {lookFrom: [0.5,2.5,-8], lookAt: [0,1,0], directionUp: [0,1,0], fieldOfView: 45, aspectRatio: [4,3],
               aperture: 0.05, distanceToFocus: 8, startTime: 0, endTime: 1},
      background: {type: "gradient", direction: [0,1,0], colour0: [0.9,0.9,1.0], colour1: [0.3,0.5,0.9]},
      textures: {r: {type: "solidColour", colour: [0.8,0.2,0.2]}, w: {type: "solidColour", colour: [0.9,0.9,0.9]},
                 k: {type: "solidColour", colour: [0.1,0.1,0.1]}, m: {type: "solidColour", colour: [0.7,0.7,0.8]}},
      materials: {red: {type: "lambertian", texture: "r"}, white: {type: "lambertian", texture: "w"},
                  black: {type: "lambertian", texture: "k"}, mirror: {type: "metal", albedo: [0.9,0.8,0.7], fuzz: 0.1},
                  glass: {type: "dielectric", indexOfRefraction: 1.5}, mist: {type: "isotropic", albedo: "m"}, smoke: {type: "isotropic", albedo: "k"},
                  tiles: {type: "checkered", odd: "white", even: "black", tileDensity: 4},
                  check: {type: "checkered", odd: "tiles", even: "red", tileDensity: 2}},
      objects: {%s},
      world: [%s],
    }""" % (objs, ", ".join('"%s"' % w for w in world))


@pytest.mark.parametrize("with_mesh,extra_spheres,nested,gpu_build", [(False, 0, False, False), (False, 14, False, False), (False, 150, False, False),
                                                                     (True, 0, False, False), (True, 150, False, False), (False, 0, True, False),
                                                                     (True, 150, True, False), (True, 0, False, True), (True, 150, True, True)])
def test_every_object_kind_against_the_oracle(rt, orc, tmp_path, monkeypatch, with_mesh, extra_spheres, nested, gpu_build):
    # (False, 0): a world of 7 leaves is swept; (False, 14): own-box tree staged in LDS; (False, 150): too large
    # for LDS - four-wide world tree, k_extend_refill with media; (True, ..): meshes and media together - the
    # general four-wide kernel, a medium bounded by a mesh; nested: media bounded by media (plain and behind
    # transforms), which only the general four-wide kernel answers - a small world takes it too then; gpu_build: the
    # four-triangle mesh's tree comes from the GPU builder (csrc/device/bvh_build.hip: one fold level, leaves only)
    if gpu_build:
        monkeypatch.setenv("RT_GPU_BUILD", "1")
    text = _corners_scene(str(tmp_path), with_mesh, extra_spheres, nested)
    scene = rt.Scene.from_text(text)
    o = orc.OracleScene.from_text(text)
    r = rt.Renderer(0)
    r.upload(scene)
    assert r.upload_stats().gpu_built_trees == (1 if gpu_build else 0)
    W = 96
    H = scene.image_height(W)
    assert H == o.image_height(W)
    prim = o.primary_rays(W, H, 0, H, 0, 11)
    ph = o.trace(prim, 11)
    check_hits(r.trace(prim, 11), ph, ph["second_t"])
    rng = np.random.default_rng(5)
    ok = ph["hit"] == 1
    n = int(ok.sum())
    rays = np.zeros(2 * n, dtype=prim.dtype)
    rays["o"] = np.tile(ph["p"][ok], (2, 1))
    d = rng.normal(size=(2 * n, 3))
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["time"] = rng.uniform(0, 1, 2 * n)
    rays["tmin"] = 0.001
    rays["tmax"] = np.inf
    want = o.trace(rays, 12)
    check_hits(r.trace(rays, 12), want, want["second_t"])
    rgb8, sums, st = r.render_rows(W, H, 0, H, 8, 20, seed=4)
    o8, osums, orays = o.render(W, H, 0, H, 8, 20, seed=4, threads=8)
    assert_image_parity(sums, osums, st.rays, orays, 8, 20, label="every object kind",
                        noise=lambda: o.render(W, H, 0, H, 8, 20, seed=5, threads=8)[1])


def test_a_world_of_one_mesh(rt, orc, gpu_scenes):
    # One boxed object in the world: BVHNode::from_vec makes a plain list of it, no world tree (src/hitting.rs:165-170) - the
    # knight alone on its board.  The mesh must still be walked through its four-wide tree (round 3: it fell back to the
    # binary walk), and rays scattered far along the unbounded board must still be brought near for the f32 box tests
    # (the bounding sphere comes from the mesh's root box then, not from a world tree).
    gpu_scenes("knight")  # (makes sure examples/knight.obj exists)
    import re
    text = open(os.path.join(ROOT, "examples", "knight.json5")).read()
    text, n = re.subn(r"world:\s*\[[^\]]*\]", 'world: ["board", "knight"]', text)
    assert n == 1
    scene = rt.Scene.from_text(text, ROOT)
    o = orc.OracleScene.from_text(text, ROOT)
    r = rt.Renderer(0)
    r.upload(scene)
    assert r.upload_stats().n_nodes4 > 1000
    W = 64
    H = scene.image_height(W)
    prim = o.primary_rays(W, H, 0, H, 0, 11)
    ph = o.trace(prim, 11)
    check_hits(r.trace(prim, 11), ph, ph["second_t"])
    on_mesh = (ph["hit"] == 1) & (ph["sub"] > 0)  # (a triangle of the mesh: its index within the mesh is the part)
    targets = ph["p"][ph["hit"] == 1]
    rng = np.random.default_rng(6)
    parts = []
    for dist in [1e3, 1e5, 1e7]:
        n_t = len(targets)
        d = rng.normal(size=(n_t, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.zeros(n_t, dtype=prim.dtype)
        rays["o"] = targets + rng.normal(scale=1e-3, size=(n_t, 3)) - d * dist
        rays["d"] = d
        rays["time"] = rng.uniform(0, 1, n_t)
        rays["tmin"] = 0.001
        rays["tmax"] = np.inf
        parts.append(rays)
    far = np.concatenate(parts)
    want, got = o.trace(far, 9), r.trace(far, 9)
    assert np.array_equal(got["hit"], want["hit"])
    hit = want["hit"] == 1
    tie = np.abs(want["second_t"] - want["t"]) <= 1e-9 * np.abs(want["t"])
    same = (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    assert np.all(same[hit & ~tie]) and on_mesh.any()
    rgb8, sums, st = r.render_rows(W, H, 0, H, 8, 20, seed=4)
    o8, osums, orays = o.render(W, H, 0, H, 8, 20, seed=4, threads=8)
    assert_image_parity(sums, osums, st.rays, orays, 8, 20, label="one mesh",
                        noise=lambda: o.render(W, H, 0, H, 8, 20, seed=5, threads=8)[1])


def _mesh_scene(obj_path):
    return """{
      camera: {lookFrom: [0.3,2.5,-7], lookAt: [0,0.6,0], directionUp: [0,1,0], fieldOfView: 40, aspectRatio: [1,1],
               aperture: 0, distanceToFocus: 7, startTime: 0, endTime: 1},
      background: {type: "gradient", direction: [0,1,0], colour0: [1,1,1], colour1: [0.4,0.6,1.0]},
      textures: {r: {type: "solidColour", colour: [0.8,0.3,0.2]}},
      materials: {red: {type: "lambertian", texture: "r"}},
      objects: {shape: {type: "mesh", filename: "%s", material: "red"},
                turned: {type: "rotateY", prototype: "shape", degrees: 20},
                dot: {type: "sphere", centre: [2.2,0.4,0.5], radius: 0.4, material: "red"}},
      world: ["turned", "dot"],
    }""" % obj_path  # (two objects: a world of one has no world tree, and then no four-wide trees at all)


@pytest.mark.parametrize("kind", ["one", "two", "duplicates", "sliver_grid", "far_apart"])
def test_gpu_built_trees_of_awkward_meshes(rt, tmp_path, monkeypatch, kind):
    # The GPU builder (csrc/device/bvh_build.hip) on meshes that stress its corners: a single leaf (super-root only), two
    # leaves, many identical triangles (equal Morton codes: the radix tree falls back on the leaves' positions), a grid of
    # slivers whose centroids lie on a line, two clusters a long way apart (most of the code's bits unused).  The tree only
    # culls: hit records and images must equal the host-built tree's bit for bit.
    rng = np.random.default_rng(3)
    verts, faces = [], []

    def tri(a, b, c):
        verts.extend([a, b, c])
        n = len(verts)
        faces.append((n - 2, n - 1, n))

    if kind == "one":
        tri((-1, 0, 0), (1, 0, 0), (0, 1.5, 0.2))
    elif kind == "two":
        tri((-1, 0, 0), (1, 0, 0), (0, 1.5, 0.2))
        tri((-1, 0, 0.5), (0, 1.5, 0.7), (1, 0, 0.5))
    elif kind == "duplicates":
        for _ in range(37):
            tri((-1, 0, 0), (1, 0, 0), (0, 1.5, 0.2))
        tri((-1.2, 0.1, 0.4), (0.9, 0.2, 0.5), (0.1, 1.2, 0.6))
    elif kind == "sliver_grid":
        for k in range(300):
            x = -1.5 + 0.01 * k
            tri((x, 0, 0.001 * k), (x + 0.009, 0, 0.001 * k), (x, 1.2, 0.001 * k))
    else:
        for centre in ((-1.0, 0.5, 0.0), (900.0, 0.5, 4000.0)):
            for _ in range(64):
                p = np.array(centre) + rng.uniform(-0.4, 0.4, 3)
                tri(tuple(p), tuple(p + rng.uniform(-0.3, 0.3, 3)), tuple(p + rng.uniform(-0.3, 0.3, 3)))
    obj = tmp_path / "awkward.obj"
    with open(obj, "w") as f:
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % v)
        for a, b, c in faces:
            f.write("f %d %d %d\n" % (a, b, c))
    scene = rt.Scene.from_text(_mesh_scene(str(obj)))
    host = rt.Renderer(0)
    host.upload(scene)
    monkeypatch.setenv("RT_GPU_BUILD", "1")
    gpu = rt.Renderer(0)
    gpu.upload(scene)
    assert host.upload_stats().gpu_built_trees == 0 and gpu.upload_stats().gpu_built_trees == 1
    W = 64
    H = scene.image_height(W)
    rays = host.primary_rays(W, H, 0, H, 0, seed=9)
    a, b = host.trace(rays, seed=2), gpu.trace(rays, seed=2)
    for field in ("hit", "owner", "sub", "front_face", "material", "t", "p", "n", "u", "v", "second_t"):
        assert np.array_equal(a[field], b[field]), field
    assert a["hit"].sum() > 0
    ra, rb = host.render_rows(W, H, 0, H, 4, 8, seed=1), gpu.render_rows(W, H, 0, H, 4, 8, seed=1)
    assert np.array_equal(ra[1], rb[1]) and ra[2].rays == rb[2].rays


@pytest.mark.parametrize("name", ["knight", "blocks_ball", "blue_ball"])
def test_rays_that_start_far_outside_the_scene(gpu_scenes, oracle_scenes, name):
    # A path scattered off the unbounded Plane can start thousands of scene sizes away.  The f32 box tests of the tree
    # walks round the ray's origin to f32; their padding covers that only near the scene, so far origins are moved along
    # the ray to the scene's bounding sphere for the box tests (kernels.hip, far_shift): boxes may only err towards
    # "hit", the exact f64 primitive tests decide, and the hit must be the oracle's at any distance.
    sc, r = gpu_scenes(name)
    o = oracle_scenes(name)
    W = 64
    H = sc.image_height(W)
    prim = r.primary_rays(W, H, 0, H, sample=0, seed=5)
    hits = o.trace(prim, 5)
    ok = hits["hit"] == 1
    targets = hits["p"][ok]                      # points on the scene's surfaces
    rng = np.random.default_rng(4)
    rays_all = []
    for dist in [1e2, 1e4, 1e6, 1e8]:
        n = len(targets)
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.zeros(n, dtype=prim.dtype)
        jitter = rng.normal(scale=1e-3, size=(n, 3))   # some rays graze edges, some miss by a hair
        rays["o"] = targets + jitter - d * dist
        rays["d"] = d
        rays["time"] = rng.uniform(0, 1, n)
        rays["tmin"] = 0.001
        rays["tmax"] = np.inf
        rays_all.append(rays)
    rays = np.concatenate(rays_all)
    want = o.trace(rays, 9)
    got = r.trace(rays, 9)
    assert want["hit"].mean() > 0.5
    # at 1e8 scene units the oracle's own f64 hit point is only good to ~1e-8: compare what does not depend on it
    assert np.array_equal(got["hit"], want["hit"]), f"{(got['hit'] != want['hit']).sum()} of {len(rays)} far rays: hit / miss differs"
    hit = want["hit"] == 1
    tie = np.abs(want["second_t"] - want["t"]) <= 1e-9 * np.abs(want["t"])
    same = (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    assert np.all(same[hit & ~tie])
    # (a chain of transforms is one rigid map on the GPU and one re-normalisation per level in the reference: last bits)
    assert np.allclose(got["t"][hit & same], want["t"][hit & same], rtol=1e-9, atol=0)


@pytest.mark.parametrize("name", SCENES)
def test_fast_mode_hit_ids_against_the_oracle(gpu_scenes, rt, name):
    # RT_PRECISION_F32 is the declared fast mode: the same kernels in f32, compared statistically only.  SURVEY.md section 7 asks for
    # its ID-mismatch rate to be measured and reported (profiles/r03/f32_mode.json): this bounds it.  On the golden rays
    # (the oracle's f64 records) rays that are not exact ties may name another primitive only where f32 cannot tell two candidates
    # apart; measured (scripts/measure_f32.py): no ID mismatch and no hit / miss flip on any of the six scenes, on the golden rays and on a
    # 160-px frame of primary rays alike; t within 2e-3 relative at worst, 1e-5 at the 99th percentile (far hits at f32 resolution).
    # (Before round 3 the mist of blocks_ball vanished for 6 % of the rays in f32: the medium's `+ 0.0001` is below f32 resolution.)
    g = np.load(os.path.join(GOLDEN, name + "_rays.npz"))
    _, r32 = gpu_scenes(name, precision=rt.PRECISION_F32)
    got = r32.trace(g["rays"], seed=int(g["seed"]))
    want = g["hits"]
    hit = want["hit"] == 1
    assert (got["hit"] != want["hit"]).mean() <= 0.001
    both = hit & (got["hit"] == 1)
    tie = tie_mask(want, want["second_t"])
    same = (got["owner"] == want["owner"]) & (got["sub"] == want["sub"])
    judged = both & ~tie
    assert (judged & ~same).sum() <= 0.002 * judged.sum(), f"{(judged & ~same).sum()} of {judged.sum()} IDs differ in f32"
    ok = both & same
    t_err = np.abs(got["t"][ok] - want["t"][ok]) / np.maximum(np.abs(want["t"][ok]), 1e-300)
    assert np.quantile(t_err, 0.99) <= 1e-4 and t_err.max() <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("name, kind, rects", [("cornell", 1, 18), ("globe", 1, 0), ("two_spheres", 2, 0), ("blue_ball", 3, 0), ("blocks_ball", 0, 0), ("knight", 0, 0)])
def test_which_walk_a_scene_gets(rt, gpu_scenes, name, kind, rects):
    # the kernel family chosen at upload (rt_upload_stats.walk): the cheap sweep kernels are what the measured numbers rest on -
    # a scene that silently fell back to the general one would still render the same image
    _, r = gpu_scenes(name)
    walk = r.upload_stats().walk
    assert walk & 0xFF == kind
    assert walk >> 8 == rects  # cornell's 18 rects are listed one by one for the lane groups of the tail launch
