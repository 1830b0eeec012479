#!/usr/bin/env python3
"""bench.py - whole-job path-tracing throughput on N MI355X of one node.

A "step" is one complete render of the workload (BASELINE.json configs[1]: examples/cornell.json5, 600 px wide, 100 samples per
pixel, 50 bounces) through the C ABI of librt_mi355x.so: camera rays -> [extend -> shade/compact] x bounces -> per-pixel f64 sums
-> reduce to rank 0 (RCCL, N > 1) -> 8-bit rows on rank 0's GPU.  The scene is resident in HBM before the timed region.  N > 1 is
STRONG scaling by default (the image's 100 sample indices are split over the ranks; `--weak` gives every rank 100 of its own);
`value` = closest-hit queries ("rays") of all ranks / the time of the slowest rank.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  The timed region is EXACTLY K steps on the product's own schedule; one extra step after it carries
HIP events around every launch (one lane), and `roofline` prices every kernel that is >= 10 % of that step: its VALU issue cycles
(instruction counts by class from the committed rocprofv3 --pmc passes, profiles/r*/pmc.json, x the cycles per class measured by
scripts/micro/valu_mix.hip) and its compulsory HBM bytes against the launch durations measured live.  `cpu_baseline` times the CPU
oracle (a C++ restatement of the reference's renderer - the Rust reference itself cannot be built here) on a bounded sample of the
same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (spec)
FP32_VALU_PEAK_TFLOPS = 157.3

# Bytes of one record of each flattened array as laid out in HBM (rt_device.h), f64 / f32.
REC = {
    "f64": {"ray": 60, "hit": 52, "node": 64, "object": 32, "xform": 96, "rect": 64, "sphere": 48, "triangle": 112, "plane": 128},
    "f32": {"ray": 32, "hit": 28, "node": 32, "object": 32, "xform": 48, "rect": 48, "sphere": 32, "triangle": 64, "plane": 64},
}
PRIM_KIND = {"cornell": "rect", "blocks_ball": "rect", "blue_ball": "rect", "globe": "rect", "two_spheres": "sphere", "knight": "triangle"}
# flops per box test / primitive test: slab 21, rect 17, sphere 25, triangle 52
PRIM_FLOPS = {"rect": 17, "sphere": 25, "triangle": 52, "plane": 52}


def extend_bytes_per_ray(scene_name: str, prec: str) -> float:
    """Algorithmic HBM bytes one closest-hit query moves in the extend kernel (DESIGN.md,
    "Kernels"): the ray read, the hit record written, and every node / object / transform /
    primitive record the reference traversal touches (means counted by the oracle on the same
    deterministic BVH, tests/golden/traversal_stats.json)."""
    with open(os.path.join(ROOT, "tests", "golden", "traversal_stats.json")) as f:
        st = json.load(f)["scenes"][scene_name]
    r = REC[prec]
    prim = r[PRIM_KIND.get(scene_name, "rect")]
    return (r["ray"] + r["hit"] + st["nodes_per_ray"] * r["node"] + st["leaves_per_ray"] * (r["node"] + r["object"])
            + st["xforms_per_ray"] * r["xform"] + st["prims_per_ray"] * prim)


def extend_flops_per_ray(scene_name: str) -> float:
    with open(os.path.join(ROOT, "tests", "golden", "traversal_stats.json")) as f:
        st = json.load(f)["scenes"][scene_name]
    return st["nodes_per_ray"] * 21 + st["prims_per_ray"] * PRIM_FLOPS[PRIM_KIND.get(scene_name, "rect")] + st["xforms_per_ray"] * 40


def newest_profile(name: str):
    """The newest profiles/r*/<name> (a JSON file committed with the round's rocprofv3 passes), or None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
        with open(path) as f:
            return json.load(f), os.path.relpath(path, ROOT)
    return None, None


def host_threads() -> int:
    """The host hardware threads this job may use: the affinity mask, capped by the box's own share (the GPU box
    exports OMP_NUM_THREADS = its CPU share per GPU, 16; os.cpu_count() there shows the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(scene_path: str, width: int, bounces: int, seed: int, asset_root: str = ROOT) -> dict:
    """CPU oracle on a bounded sample of the workload: same scene / width / bounces, fewer samples per pixel
    (rays per sample are the same in expectation); every host thread we may use, median of 3 runs."""
    from oracle import binding as orc

    threads = host_threads()
    sc = orc.OracleScene.load(scene_path, asset_root)
    H = sc.image_height(width)
    t0 = time.time()
    sc.render(width, H, 0, H, 1, bounces, seed, threads=threads)
    dt = time.time() - t0
    # Each of the 3 runs is about 7 s of wall time (20 s in all), at most 128 spp.
    spp2 = int(max(1, min(128, round(7.0 / max(dt, 1e-3)))))
    runs = []
    for k in range(3):
        t0 = time.time()
        _, _, rays = sc.render(width, H, 0, H, spp2, bounces, seed + k, threads=threads)
        runs.append((rays / (time.time() - t0) / 1e6, rays, time.time() - t0))
    runs.sort()
    value, rays, dt = runs[1]
    return {"value": value, "unit": "Mrays/s", "cores": threads, "kind": "port", "runs_Mrays_s": [round(r[0], 2) for r in runs],
            "sample": f"{os.path.basename(scene_path)} {width}x{H}, {spp2} spp of the workload's samples, {bounces} bounces, "
                      f"median of 3 runs ({rays} rays in {dt:.2f} s); C++ f64 restatement of the reference renderer (oracle/, "
                      f"g++ -O3 -march=x86-64-v3: built before it travels to the box, so not -march=native), {threads} threads = every host thread of the box's share"}


class MergedStats:
    """rt_stats of the concurrent batches of one step: counts and per-stream launch times added up (the streams run at
    the same time, so the stage times add up to more than the step takes)."""

    def __init__(self, parts):
        for f in ("rays", "paths", "rays_tail", "paths_tail", "paths_first", "ms_extend", "ms_shade", "ms_tail", "extend_launches", "shade_launches", "tail_launches"):
            setattr(self, f, sum(getattr(p, f) for p in parts))
        self.ms_total = max(p.ms_total for p in parts)
        self.extend_kernel = parts[-1].extend_kernel
        self.max_depth_reached = max(p.max_depth_reached for p in parts)
        self.batches = sum(p.batches for p in parts)


# ---- the VALU-issue roof.  A gfx950 SIMD executes 32 lanes of 32-bit or 16 lanes of 64-bit arithmetic a cycle: a wave64 instruction
# holds it for 2 cycles (32-bit add / mul / fma / logic / move), 4 (every f64 operation, 64-bit integer, conversions, 32-bit multiplies,
# v_cndmask, v_readlane), 8 (32-bit transcendental) or 16 (v_rcp_f64 / v_rsq_f64 / v_sqrt_f64).  Measured per class on the box by
# scripts/micro/valu_mix.hip (profiles/r03/valu_mix.txt: 4.2 / 2.3 / 8.1 / 16.2 with loop overhead) together with the SQ_INSTS_VALU_*
# counter each class lands in (valu_mix_counters.txt).  A kernel's minimum issue time is the sum over its instructions of those
# cycles; the counters resolve most classes, the rest ("other": moves and selects at 2 - 4 cycles, f64 compares / min / max /
# ldexp / div_scale / div_fixup at 4) is priced at 2 cycles for `low` and 4 for `high`.
SIMDS = 1024
SPEC_CLOCK_GHZ = 2.4
ISSUE_PEAK = SIMDS * SPEC_CLOCK_GHZ   # G SIMD issue cycles per second


def issue_cycles(c: dict):
    """(low, high) SIMD issue cycles of a kernel from its SQ_INSTS_VALU_* counters, and the instruction mix."""
    g = lambda k: float(c.get(k, 0.0))
    valu = g("SQ_INSTS_VALU")
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    t64 = g("SQ_INSTS_VALU_TRANS_F64")
    f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
    t32 = g("SQ_INSTS_VALU_TRANS_F32")
    i32, i64, cvt = g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_VALU_INT64"), g("SQ_INSTS_VALU_CVT")
    logic = max(0.0, g("SQ_INSTS_VALU_IOPS") - i32 - 2.0 * i64)   # IOPS counts a 64-bit multiply-add twice, xor / shifts once
    other = max(0.0, valu - f64 - t64 - f32 - t32 - i32 - i64 - cvt - logic)
    fixed = 4.0 * f64 + 16.0 * t64 + 8.0 * t32 + 4.0 * i64 + 4.0 * cvt
    low = fixed + 2.0 * (f32 + i32 + logic + other)
    high = fixed + 4.0 * (f32 + i32 + logic + other)
    mix = {"f64_add_mul_fma": f64, "f64_transcendental": t64, "f32_add_mul_fma": f32, "f32_transcendental": t32, "int32": i32,
           "int64": i64, "convert": cvt, "logic": logic, "other": other}
    return low, high, {k: (v / valu if valu else 0.0) for k, v in mix.items()}


def price_kernel(name: str, pk: dict, ms: float, launches: int, compulsory_bytes: float, f64_clock: float, rays: float = None):
    """One entry of roofline.kernels: the kernel's launches of the instrumented step (ms, measured live with HIP events on the
    library's stream) against its instruction counts / HBM traffic from the committed PMC passes of the same workload."""
    c = pk["counters"]
    rays = rays if rays is not None else pk["rays"]   # (the committed pass and this run trace the same rays: same seed)
    low, high, mix = issue_cycles(c)
    sec = ms * 1e-3
    out = {"kernel": name, "launches": launches, "ms": ms, "avg_launch_ms": ms / max(launches, 1), "rays": rays,
           "valu_insts_per_ray": c.get("SQ_INSTS_VALU", 0.0) / rays if rays else None,
           "issue_cycles_per_ray": [low / rays, high / rays] if rays else None, "instruction_mix": {k: round(v, 4) for k, v in mix.items()}}
    if sec > 0:
        out["issue_Gcycles_per_s"] = [low / sec / 1e9, high / sec / 1e9]
        out["issue_frac"] = [low / sec / 1e9 / ISSUE_PEAK, high / sec / 1e9 / ISSUE_PEAK]
        out["issue_frac_at_f64_clock"] = [low / sec / 1e9 / (SIMDS * f64_clock), high / sec / 1e9 / (SIMDS * f64_clock)]
        out["hbm_compulsory_bytes"] = compulsory_bytes
        out["hbm_GBps"] = compulsory_bytes / sec / 1e9
        out["hbm_frac"] = compulsory_bytes / sec / 1e9 / HBM_PEAK_GBS
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950: doubled (MI355X_MICROARCH.md, HBM;
        # calibrated on these queues' access widths in round 1, profiles/r01/traffic_calibration.txt)
        traffic = c["FETCH_SIZE"] * 1024.0 * 2.0 + c["WRITE_SIZE"] * 1024.0
        out["traffic_bytes"] = traffic
        out["traffic_over_compulsory"] = traffic / compulsory_bytes if compulsory_bytes else None
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        out["valu_lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])   # share of the 64 lanes a VALU instruction has on
    if c.get("SQ_WAVE_CYCLES"):
        out["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        out["issue_stall_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_BUSY_CYCLES"):
            # resident waves per SIMD while the kernel ran: wave-cycles / (SQ busy cycles x 4 SIMDs) (both in quad-cycles, summed over the chip)
            out["waves_per_simd"] = c["SQ_WAVE_CYCLES"] / (c["SQ_BUSY_CYCLES"] * 4.0)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("GRBM_GUI_ACTIVE") and pk.get("duration_ms"):
        out["effective_clock_GHz_in_pmc_pass"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (pk["duration_ms"] * 1e-3) / 1e9
    for k in ("vgpr", "sgpr", "scratch_bytes_per_lane", "lds_bytes"):
        if pk.get(k) is not None:
            out[k] = pk[k]
    return out


def load_scene(rt, args, scene_name):
    scene_path = os.path.join(ROOT, "examples", scene_name + ".json5")
    asset_root = ROOT
    if scene_name == "knight":
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import gen_knight

        if args.triangles > 0:  # another size of the stand-in, one copy per rank, outside the tree
            import shutil
            import tempfile

            asset_root = tempfile.mkdtemp(prefix="knight_")
            os.makedirs(os.path.join(asset_root, "examples"))
            gen_knight.ensure(os.path.join(asset_root, "examples", "knight.obj"), args.triangles)
            shutil.copy(scene_path, os.path.join(asset_root, "examples", "knight.json5"))
            scene_path = os.path.join(asset_root, "examples", "knight.json5")
        else:
            gen_knight.ensure(os.path.join(ROOT, "examples", "knight.obj"))
    t0 = time.perf_counter()
    scene = rt.Scene.load(scene_path, asset_root)
    return scene, scene_path, asset_root, (time.perf_counter() - t0) * 1e3


def run_workload(args, scene_name, steps, warmup, torch, rt, parallel, dist, rank, local_rank, world, with_cpu):
    """Times `steps` whole images of one scene; returns the JSON line (rank 0) or None."""
    scene, scene_path, asset_root, ms_load = load_scene(rt, args, scene_name)
    W = args.width
    H = scene.image_height(W)
    npix = W * H
    K = max(1, args.overlap)
    renderers = []
    for _ in range(K):
        r = rt.Renderer(device=local_rank, precision=rt.PRECISION_F64 if args.precision == "f64" else rt.PRECISION_F32,
                        max_paths=(1 << 26) // K, lanes=args.lanes)
        r.upload(scene)
        renderers.append(r)
    renderer = renderers[0]
    up = renderer.upload_stats()

    dev = torch.device("cuda", local_rank)
    accum = torch.zeros(npix * 3, dtype=torch.float64, device=dev)
    extra_accum = [torch.zeros(npix * 3, dtype=torch.float64, device=dev) for _ in range(K - 1)]
    rgb8 = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
    strong = args.strong
    total_spp = args.spp if strong else args.spp * world
    s0, ns = parallel.split_samples(total_spp, world, rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(collect):
        accum.zero_()
        torch.cuda.synchronize()  # torch's stream -> the library's stream hand-off
        if K == 1:
            st = renderer.render_accum_device(W, H, 0, H, s0, ns, args.bounces, args.seed, accum.data_ptr(), want_stats=True)
        else:
            st = MergedStats(parallel.render_concurrent(renderers, [accum] + extra_accum, W, H, 0, H, s0, ns, args.bounces, args.seed))
        parallel.reduce_to_root(accum, dist)
        if rank == 0:
            torch.cuda.synchronize()
            renderer.quantize_device(accum.data_ptr(), npix, total_spp, rgb8.data_ptr())
        if collect is not None:
            collect.append(st)

    if dist is not None:  # RCCL sets its connections up in the first collective of each kind: not a step, never timed
        parallel.reduce_to_root(torch.zeros(1024, dtype=torch.float64, device=dev), dist)
    for _ in range(warmup):
        step(None)
    stats = []
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):   # EXACTLY `steps` steps, all on the product's own schedule (no events, the library's lanes)
        step(stats)
    barrier()
    dt = time.perf_counter() - t0
    # One more step, OUTSIDE the timed region, with HIP events around every launch on the library's stream: the events cost
    # the stream ~5 us each and the library runs ONE lane while they are on (a launch's duration is then its own, not
    # stretched by the other lane's kernels).  Its launch durations feed `roofline`; `value` is the steps above alone.
    instrumented = []
    if not args.no_launch_timing:
        for r in renderers:
            r.set_launch_timing(True)
        step(instrumented)
        torch.cuda.synchronize()
        for r in renderers:
            r.set_launch_timing(False)

    rays = sum(s.rays for s in stats)
    last = instrumented[-1] if instrumented else stats[-1]
    t = torch.tensor([dt, float(rays), sum(s.ms_total for s in stats)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        rays_all = float(tsum[1])
    else:
        rays_all = float(rays)
    if rank != 0:
        return None
    ms_per_step = dt / steps * 1e3

    # ---- what the kernels of one step move and execute
    r = REC[args.precision]
    elem = 8 if args.precision == "f64" else 4
    counts = scene.counts()
    # Records the walks can touch (everything but materials / texels), as laid out in HBM.
    scene_bytes = (counts["nodes"] * r["node"] + counts["objects"] * r["object"] + counts["xforms"] * r["xform"]
                   + counts["rects"] * r["rect"] + counts["spheres"] * r["sphere"] + counts["triangles"] * r["triangle"]
                   + counts["planes"] * r["plane"])
    rays_step, rays_tail, paths, paths_tail = float(last.rays), float(last.rays_tail), float(last.paths), float(last.paths_tail)
    ms_extend, ms_shade, ms_tail = float(last.ms_extend), float(last.ms_shade), float(last.ms_tail)
    n_extend, n_shade, n_tail = int(last.extend_launches), int(last.shade_launches), int(last.tail_launches)
    fused = last.extend_kernel == 2
    kernel = "k_bounce" if fused else "k_extend_refill" if last.extend_kernel == 1 else "k_extend"
    walk_kernel = kernel  # (the kernel that walks the scene: SURVEY 8d's nominal flops and touched bytes are its)
    rays_nontail = rays_step - rays_tail
    # Algorithmic (compulsory) HBM bytes per kernel and step: queue entries and hit records are read /
    # written once per ray, the scene once per launch.  The records a traversal touches again and again (SURVEY.md 8d's
    # per-ray sum) are served by SGPR loads / LDS / L2 / Infinity Cache and are reported separately as `touched`.
    path_bytes = 10 * elem + 4                       # ray 7 + throughput 3 fields + path id
    hit_bytes = (4 + (2 if up.needs_uv else 0)) * elem + 4   # t, n (+ u, v where a material reads them) + material | face
    ray_bytes = 7 * elem + 4
    sample_bytes = 3 * elem
    survivors = (rays_nontail - paths) + paths_tail if not fused else rays_step - paths   # queue entries written by shade / bounce launches
    compulsory = {
        # (the first bounce of a batch on a sweep scene makes its camera rays itself: those queue entries are never read)
        "k_bounce": path_bytes * (rays_step - float(last.paths_first)) + path_bytes * survivors + sample_bytes * paths + scene_bytes * n_extend,
        "k_extend": (ray_bytes + hit_bytes) * rays_nontail + scene_bytes * n_extend,
        "k_extend_refill": (ray_bytes + hit_bytes) * rays_nontail + scene_bytes * n_extend,
        "k_shade": (path_bytes + hit_bytes) * rays_nontail + path_bytes * survivors + sample_bytes * (paths - paths_tail),
        "k_bounce_tail": path_bytes * paths_tail + sample_bytes * paths_tail + scene_bytes * n_tail,
    }
    timed = [(kernel, ms_extend, n_extend)] + ([("k_shade", ms_shade, n_shade), ("k_bounce_tail", ms_tail, n_tail)] if not fused else [])
    ms_launch_sum = sum(ms for _, ms, _ in timed)
    rays_per_launch = rays_nontail / max(n_extend, 1) if not fused else rays_step / max(n_extend, 1)
    ms_per_launch = ms_extend / max(n_extend, 1)
    bytes_per_launch = compulsory[kernel] / max(n_extend, 1)
    achieved_gbs = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    touched_bpr = extend_bytes_per_ray(scene_name, args.precision)
    touched_gbs = touched_bpr * rays_nontail / (ms_extend * 1e-3) / 1e9 if ms_extend > 0 and not fused else \
        touched_bpr * rays_step / (ms_extend * 1e-3) / 1e9 if ms_extend > 0 else 0.0
    flops = extend_flops_per_ray(scene_name) * (rays_step if fused else rays_nontail) / (ms_extend * 1e-3) / 1e12 if ms_extend > 0 else 0.0
    valu_peak = FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else FP32_VALU_PEAK_TFLOPS
    pmc_all, pmc_path = newest_profile("pmc.json")
    mix_cal, mix_path = newest_profile("valu_mix.json")
    f64_clock = mix_cal["f64_clock_GHz"] if mix_cal else SPEC_CLOCK_GHZ
    pmc = (pmc_all or {}).get(scene_name)
    # the committed passes describe the default workload of a scene (default mesh, 100 spp, f64, this bounce limit)
    if pmc and not (args.triangles == 0 and pmc.get("precision") == args.precision and abs(pmc["rays_per_step"] - rays_step) <= 1e-6 * rays_step):
        pmc = None
    hbm = {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
           "bytes_per_launch": bytes_per_launch, "bytes_per_ray": compulsory[kernel] / max(rays_step if fused else rays_nontail, 1.0),
           "scene_bytes": scene_bytes, "traffic": None,
           "touched": {"bytes_per_ray": touched_bpr, "achieved_GBps": touched_gbs,
                       "note": "all records the reference traversal touches per ray (SURVEY 8d sum, f64 sizes); served by SGPR loads / LDS / L2 / "
                               "Infinity Cache, NOT HBM bytes (BASELINE.md)"}}
    priced = []
    if pmc and instrumented:
        for name, ms, n in timed:
            if name in pmc["kernels"] and n > 0:
                priced.append(price_kernel(name, pmc["kernels"][name], ms, n, compulsory[name], f64_clock))
        for e in priced:
            e["share_of_instrumented_step"] = e["ms"] / float(last.ms_total) if last.ms_total else None
    # the top-level fields describe the kernel that takes the largest part of the step (two_spheres: k_shade, not k_extend)
    dom = max((e for e in priced if "issue_frac" in e), key=lambda e: e["ms"], default=None)
    if dom is not None:
        kernel, n_extend = dom["kernel"], dom["launches"]
        ms_per_launch = dom["avg_launch_ms"]
        rays_per_launch = dom["rays"] / max(n_extend, 1)
        hbm.update({"achieved": dom["hbm_GBps"], "frac": dom["hbm_frac"], "bytes_per_launch": dom["hbm_compulsory_bytes"] / max(n_extend, 1),
                    "bytes_per_ray": dom["hbm_compulsory_bytes"] / max(dom["rays"], 1.0)})
        hbm["traffic"] = dom.get("traffic_bytes", 0.0) / max(n_extend, 1) if "traffic_bytes" in dom else None
        # every kernel of the step, launches and all (raygen, accumulate, scans included), against the step's wall time
        whole_low = sum(issue_cycles(k["counters"])[0] for k in pmc["kernels"].values())
        whole_high = sum(issue_cycles(k["counters"])[1] for k in pmc["kernels"].values())
        roofline = {"bound": "valu_issue", "kernel": kernel, "achieved": dom["issue_Gcycles_per_s"][0], "peak": ISSUE_PEAK,
                    "unit": "G SIMD issue cycles/s", "frac": dom["issue_frac"][0], "frac_range": dom["issue_frac"],
                    "frac_at_f64_clock": dom["issue_frac_at_f64_clock"], "f64_clock_GHz": f64_clock, "spec_clock_GHz": SPEC_CLOCK_GHZ,
                    "traffic": hbm["traffic"],
                    "whole_step": {"issue_frac": [whole_low / (ms_per_step * 1e-3) / 1e9 / ISSUE_PEAK, whole_high / (ms_per_step * 1e-3) / 1e9 / ISSUE_PEAK],
                                   "note": "issue cycles of EVERY kernel of a step (PMC) over ms_per_step of the timed region (the product's two-lane schedule)"},
                    "sources": {"pmc": pmc_path, "instruction_class_cycles": mix_path},
                    "note": "achieved = the SIMD cycles the kernel's VALU instructions need at the least (instruction counts by class from the committed "
                            "rocprofv3 --pmc passes of this workload x cycles per class: 2 for 32-bit add / mul / fma / logic / moves, 4 for f64 and 64-bit "
                            "integer operations and conversions, 8 / 16 for f32 / f64 rcp, rsq, sqrt; instructions no class counter names are priced at 2 "
                            "cycles for frac_range[0] = frac and at 4 for frac_range[1]) per launch / the launch's duration measured here; peak = 1024 SIMDs x "
                            "the 2.4 GHz specification clock (frac_at_f64_clock: x the clock the chip holds under f64 arithmetic, measured). The path is bound "
                            "by VALU issue and, in the tree walks, by dependent-fetch latency at low lane utilisation (kernels[*].valu_lanes_active) - not "
                            "by HBM bandwidth, whose fraction is under `hbm`"}
    else:
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                    "note": "no rocprofv3 --pmc pass is committed for this workload (profiles/r*/pmc.json) or launch timing is off: priced against HBM only"}
    roofline.update({"hbm": hbm, "launches": n_extend, "avg_launch_ms": ms_per_launch, "rays_per_launch": rays_per_launch, "kernels": priced,
                     "launch_timing": "HIP events on the library's stream around every launch of ONE EXTRA step run after the timed region (one lane "
                                      "while events are on); rocprofv3 --kernel-trace --stats of `RT_ONE_LANE=1 bench.py` agrees (profiles/)"})
    out = {
        "metric": "Mrays/sec (whole node)",
        "value": rays_all / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic" if scene_name == "knight" else "reference example scene",
        "config": {
            "workload": f"examples/{scene_name}.json5 {W}x{H}, {ns} spp on this GPU ({total_spp} spp image, "
                        f"{'strong' if strong else 'weak'} scaling), {args.bounces} bounces, sample-split across {world} GPU(s), reduce to rank 0, quantise",
            "spp_per_gpu": ns, "spp_image": total_spp,
            "triangles": counts["triangles"], "paths_per_step": npix * total_spp, "rays_per_step": rays_all / steps,
            "wall_clock_per_image_s": dt / steps, "seed": args.seed, "concurrent_batches_per_gpu": K, "lanes": args.lanes,
        },
        "roofline": roofline,
        "roofline_valu": {"kernel": walk_kernel, "achieved": flops, "peak": valu_peak, "unit": "TFLOP/s", "frac": flops / valu_peak,
                          "note": "SURVEY 8d's nominal flops (21 per box, 17 rect, 25 sphere, 52 triangle): counts neither divisions as sequences nor addressing"},
        # the extra, instrumented step (one lane, events on): what the per-kernel prices are made of.  Its launches add up to
        # less than its own GPU time; ms_per_step above is the product's schedule (two lanes where they pay) and is not its mean.
        "instrumented_step": {"extend": ms_extend, "shade": ms_shade, "tail": ms_tail, "gpu_total": float(last.ms_total),
                              "launches_ms_sum": ms_launch_sum, "launches_fit_the_step": bool(ms_launch_sum <= float(last.ms_total) * 1.0005),
                              "extend_launches": n_extend, "shade_launches": n_shade, "tail_launches": n_tail,
                              "lanes": 1 if instrumented else None, "timed": bool(instrumented)},
        "stage_counts": {"rays": int(last.rays), "paths": int(last.paths), "rays_tail": int(last.rays_tail), "paths_tail": int(last.paths_tail), "paths_first": int(last.paths_first),
                         "extend_kernel": int(last.extend_kernel), "max_depth_reached": int(last.max_depth_reached), "batches": int(last.batches)},
        "scene_setup_ms": {"load_and_flatten": ms_load, "upload_total": up.ms_total, "upload_validate": up.ms_validate,
                           "upload_build_layouts": up.ms_build, "upload_copy": up.ms_copy, "device_bytes": int(up.bytes_device),
                           "note": "outside the timed region, as the reference's own timer starts after load_config (src/main.rs:84)"},
    }
    out["stage_ms_last_step"] = dict(out["instrumented_step"])   # (the name earlier rounds' records used)
    if world == 1 and with_cpu:
        out["cpu_baseline"] = cpu_baseline(scene_path, W, args.bounces, args.seed, asset_root)
    for rr in renderers:
        rr.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=600)
    ap.add_argument("--spp", type=int, default=100, help="samples per pixel of the image (strong scaling) / per GPU (--weak)")
    ap.add_argument("--bounces", type=int, default=50)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true",
                    help="skip the extra, instrumented step after the timed region (rocprofv3 --pmc passes: counters of exactly the steps asked for)")
    ap.add_argument("--no-extra", action="store_true", help="N = 1, default scene only: skip the second workload (knight) of the `workloads` array")
    ap.add_argument("--triangles", type=int, default=0,
                    help="--scene knight only: generate the stand-in mesh with about this many triangles in a scratch "
                         "directory (default: the ~100 k-triangle examples/knight.obj; BASELINE configs[4] names ~1 M)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="(round-1 experiment) render a rank's samples as this many concurrent contexts, one host thread each; "
                         "the library's own lanes (--lanes) replace it")
    ap.add_argument("--lanes", type=int, default=0, help="rt_options.lanes: streams that share a call's sample batches (0: the library's default)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --spp is the whole image's sample count, split over the GPUs (the default for N > 1)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --spp samples per GPU, an N x spp image")
    args = ap.parse_args()
    args.strong = not args.weak  # (one GPU: the same image either way - labelled like the N > 1 lines it is compared with)

    if args.overlap > 1:  # two streams share the CUs: 8 persistent blocks per CU each measure best
        os.environ.setdefault("RT_GRID_MULT", "8")

    import torch

    import raytracer_amd as rt
    from raytracer_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    out = run_workload(args, args.scene, args.steps, args.warmup, torch, rt, parallel, dist, rank, local_rank, world,
                       with_cpu=not args.no_cpu_baseline)
    # BASELINE.json's metric names cornell AND knight: the default one-GPU run carries the knight line too.
    if world == 1 and args.scene == "cornell" and not args.no_extra and args.triangles == 0 and args.spp == 100:
        extra = run_workload(args, "knight", max(2, min(args.steps, 5)), 1, torch, rt, parallel, dist, rank, local_rank, world,
                             with_cpu=not args.no_cpu_baseline)
        keys = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "data", "config", "roofline", "instrumented_step",
                "stage_counts", "scene_setup_ms", "cpu_baseline")
        out["workloads"] = [{k: out[k] for k in keys if k in out}, {k: extra[k] for k in keys if k in extra}]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
