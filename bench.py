#!/usr/bin/env python3
"""bench.py - whole-job path-tracing throughput on N MI355X of one node.

A "step" is one complete render of the workload (BASELINE.json configs[1]:
examples/cornell.json5, 600 px wide, 100 samples per pixel PER GPU, 50 bounces) through the
C ABI of librt_mi355x.so: camera rays -> [extend -> shade/compact] x bounces -> per-pixel
f64 sums -> reduce to rank 0 (RCCL, N > 1) -> 8-bit rows on rank 0's GPU.  The scene is
resident in HBM before the timed region.  Weak scaling: every rank traces its own 100
sample indices of every pixel (rank r: samples [100 r, 100 r + 100)), so N GPUs deliver an
N x 100 spp image; `value` = closest-hit queries ("rays") of all ranks / time.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (extend = BVH traversal)
against HBM bandwidth with the algorithmic bytes per ray of DESIGN.md; `cpu_baseline` times
the CPU oracle (a C++ restatement of the reference's renderer - the Rust reference itself
cannot be built here) on a bounded sample of the same workload, rank 0, N = 1 only.
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


def measured_traffic(scene_name: str, prec: str, kernel: str = None):
    """HBM bytes per k_extend launch from the PMC passes committed under profiles/ (separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950); None when no pass exists for this scene."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        with open(path) as f:
            t = json.load(f).get(scene_name)
        if t and t.get("precision") == prec and (kernel is None or t.get("kernel") == kernel):
            return t["traffic_bytes_per_launch"]
    return None


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
        for f in ("rays", "paths", "rays_tail", "ms_extend", "ms_shade", "ms_tail", "extend_launches", "shade_launches", "tail_launches"):
            setattr(self, f, sum(getattr(p, f) for p in parts))
        self.ms_total = max(p.ms_total for p in parts)
        self.extend_kernel = parts[-1].extend_kernel


def measured_valu(scene_name: str, prec: str, kernel: str):
    """VALU wave-instructions per ray of the dominant kernel from the SQ_INSTS_VALU pass committed under profiles/
    (rocprofv3 --pmc, its own run); None when no pass exists for this scene / kernel."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "valu.json")), reverse=True):
        with open(path) as f:
            t = json.load(f).get(scene_name)
        if t and t.get("precision") == prec and t.get("kernel") == kernel:
            return t
    return None


VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4 / 1e9   # G wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz


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
    for k in range(steps):
        # HIP events around every launch cost the stream ~5 us each, and the library runs ONE lane
        # while they are on (a launch's duration is then its own, not stretched by the other lane's kernels), so only
        # the LAST timed step carries them: its launch durations feed `roofline`, all steps feed `value`.
        if k == steps - 1:
            for r in renderers:
                r.set_launch_timing(True)
        step(stats)
    barrier()
    dt = time.perf_counter() - t0
    for r in renderers:
        r.set_launch_timing(False)

    rays = sum(s.rays for s in stats)
    last = stats[-1]
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

    # ---- the dominant kernel, from the launches of the last timed step (HIP events on the library's own streams)
    ms_extend, n_extend = float(last.ms_extend), float(last.extend_launches)
    r = REC[args.precision]
    counts = scene.counts()
    # Records k_extend can touch (everything but materials / texels), as laid out in HBM.
    scene_bytes = (counts["nodes"] * r["node"] + counts["objects"] * r["object"] + counts["xforms"] * r["xform"]
                   + counts["rects"] * r["rect"] + counts["spheres"] * r["sphere"] + counts["triangles"] * r["triangle"]
                   + counts["planes"] * r["plane"])
    # Algorithmic (compulsory) HBM bytes of one k_extend launch: every live ray is read once
    # (origin, direction, time, path id) and its hit record written once; the scene is read once.
    # The records a traversal touches again and again (SURVEY.md 8d's per-ray sum) are served by
    # LDS / L2 / Infinity Cache and are reported separately as `touched`.
    # (the tail launch of a two-kernel render is another kernel: its rays and time are reported apart)
    rays_step = float(last.rays)
    rays_tail = float(last.rays_tail)
    ms_tail = float(last.ms_tail)
    fused = last.extend_kernel == 2
    kernel = "k_bounce" if fused else "k_extend_refill" if last.extend_kernel == 1 else "k_extend"
    if fused and last.tail_launches == 0 and n_extend > 0:
        pass
    rays_per_launch = (rays_step - rays_tail) / max(n_extend, 1.0)
    ms_per_launch = ms_extend / max(n_extend, 1.0)
    # Small worlds of rects and spheres run extend + shade as ONE kernel per bounce (k_bounce; the
    # library reports its time as ms_extend and no shade launches): it reads a path's queue entry
    # (ray 56 B + throughput 24 B + path id 4 B in f64) and writes one for every survivor; the
    # survivors of all launches are the rays of the later launches = rays - paths.
    path_bytes = 84 if args.precision == "f64" else 44
    paths = float(last.paths)
    if fused:
        bytes_per_launch = (path_bytes * rays_step + path_bytes * (rays_step - paths)) / max(n_extend, 1.0) + scene_bytes
        bytes_per_ray = path_bytes * (2.0 - paths / max(rays_step, 1.0))
    else:
        bytes_per_launch = (r["ray"] + r["hit"]) * rays_per_launch + scene_bytes
        bytes_per_ray = r["ray"] + r["hit"]
    achieved_gbs = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    touched_bpr = extend_bytes_per_ray(scene_name, args.precision)
    touched_gbs = touched_bpr * (rays_step - rays_tail) / (ms_extend * 1e-3) / 1e9 if ms_extend > 0 else 0.0
    flops = extend_flops_per_ray(scene_name) * (rays_step - rays_tail) / (ms_extend * 1e-3) / 1e12 if ms_extend > 0 else 0.0
    valu_peak = FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else FP32_VALU_PEAK_TFLOPS
    default_mesh = args.triangles == 0
    valu = measured_valu(scene_name, args.precision, kernel) if default_mesh else None
    hbm = {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
           "bytes_per_launch": bytes_per_launch, "bytes_per_ray": bytes_per_ray, "scene_bytes": scene_bytes,
           "traffic": measured_traffic(scene_name, args.precision, kernel) if default_mesh else None,  # PMC passes exist for the default meshes only
           "touched": {"bytes_per_ray": touched_bpr, "achieved_GBps": touched_gbs,
                       "note": "all records the reference traversal touches per ray (SURVEY 8d sum, f64 sizes); served by SGPR loads / LDS / L2 / "
                               "Infinity Cache, NOT HBM bytes (BASELINE.md)"}}
    if valu is not None:
        # What binds: VALU issue.  wave-instructions per ray (SQ_INSTS_VALU of the kernel / its rays, a property of
        # kernel + scene, from the committed PMC pass) x the rays of a launch / the launch's duration measured here.
        issue = valu["valu_insts_per_ray"] * rays_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        roofline = {"bound": "valu_issue", "kernel": kernel, "achieved": issue, "peak": VALU_ISSUE_PEAK, "unit": "Gwaveinstr/s",
                    "frac": issue / VALU_ISSUE_PEAK, "traffic": hbm["traffic"],
                    "valu_insts_per_ray": valu["valu_insts_per_ray"], "valu_source": valu.get("source"),
                    "note": kernel + " is bound by VALU issue (f64: one wave64 instruction per 4 cycles per SIMD; divisions and square roots are "
                            "instruction sequences) and, on meshes, by dependent-fetch latency - not by HBM bandwidth, whose fraction is under `hbm`"}
    else:
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": hbm["traffic"],
                    "note": "no SQ_INSTS_VALU pass is committed for this scene / kernel: priced against HBM only"}
    roofline.update({"hbm": hbm, "launches": n_extend, "avg_launch_ms": ms_per_launch, "rays_per_launch": rays_per_launch,
                     "launch_timing": "HIP events on the library's stream around every launch of the last timed step (one lane while timed)"})
    out = {
        "metric": "Mrays/sec (whole node)",
        "value": rays_all / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
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
        "roofline_valu": {"kernel": kernel, "achieved": flops, "peak": valu_peak, "unit": "TFLOP/s", "frac": flops / valu_peak,
                          "note": "SURVEY 8d's nominal flops (21 per box, 17 rect, 25 sphere, 52 triangle): counts neither divisions as sequences nor addressing"},
        "stage_ms_last_step": {"extend": ms_extend, "shade": float(last.ms_shade), "tail": ms_tail, "gpu_total": float(last.ms_total),
                               "extend_launches": int(last.extend_launches), "shade_launches": int(last.shade_launches),
                               "tail_launches": int(last.tail_launches)},
        "scene_setup_ms": {"load_and_flatten": ms_load, "upload_total": up.ms_total, "upload_validate": up.ms_validate,
                           "upload_build_layouts": up.ms_build, "upload_copy": up.ms_copy, "device_bytes": int(up.bytes_device),
                           "note": "outside the timed region, as the reference's own timer starts after load_config (src/main.rs:84)"},
    }
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
    args.strong = args.strong or (args.gpus > 1 and not args.weak) or int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.weak

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
        keys = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "data", "config", "roofline", "stage_ms_last_step",
                "scene_setup_ms", "cpu_baseline")
        out["workloads"] = [{k: out[k] for k in keys if k in out}, {k: extra[k] for k in keys if k in extra}]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
