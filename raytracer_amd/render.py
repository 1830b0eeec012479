"""Whole-node render of one image: the reference's `raytracer` command over N GPUs.

    python -m raytracer_amd.render [OPTIONS] <input-file> <output-file>                      (one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m raytracer_amd.render [OPTIONS] <input-file> <output-file>                         (N GPUs)

Replaces `main()`'s row loop (src/main.rs:117-154) at node scale, with the reference's options
(`struct Opt`, src/main.rs:32-71) and its `.PART` behaviour (src/main.rs:191-246, src/part_file.rs):

* the image is rendered in bands of rows from the top of the picture down (the reference emits rows
  `.rev()`, src/main.rs:120); rows found in `--recover-from` are not rendered again;
* per band, rank r traces samples `split_samples(spp, N, r)` of every pixel into its own f64 sums on
  its GPU (`rt_render_accum_device`), ONE reduce to rank 0 (RCCL over xGMI; nothing else crosses
  GPUs), rank 0 quantises (`rt_quantize_device`: /spp, sqrt, colour_to_raw, src/main.rs:138-142,172-177);
* the rows of a band become final after its reduce: rank 0 hands `(j, row)` to the monitor thread in
  descending j, which draws the progress bar and appends one `.PART` record per row, exactly as
  `monitor_progress` does; the part file is removed when the image is complete;
* the random streams are keyed by (pixel, sample index, bounce), so the picture is the same for any N
  (up to the order of the f64 sums).

`render_image` takes the per-rank renderer as an object ("engine") so that the band / reduce / part
file logic runs under gloo on CPU in tests/test_distributed_cpu.py with the CPU oracle as the engine;
the product engine is `GpuEngine` below (no CPU fallback: without a HIP device rt_create fails).
"""
from __future__ import annotations

import argparse
import os
import queue
import sys
import threading
import time
from typing import List, Optional, Tuple

import numpy as np

from . import parallel


class GpuEngine:
    """One rank's GPU renderer behind the C ABI (include/rt_mi355x.h)."""

    def __init__(self, scene_path: str, device: int, precision: str = "f64", max_paths: int = 0, asset_root: Optional[str] = None,
                 overlap: int = 1):
        import torch

        import raytracer_amd as rt

        self.torch = torch
        self.scene = rt.Scene.load(scene_path, asset_root)
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        # overlap > 1: that many contexts (streams) on this GPU, a band's samples split over them and rendered
        # concurrently (parallel.render_concurrent)
        self.renderers = []
        for _ in range(max(1, overlap)):
            r = rt.Renderer(device=device, precision=rt.PRECISION_F64 if precision == "f64" else rt.PRECISION_F32, max_paths=max_paths)
            r.upload(self.scene)
            self.renderers.append(r)
        self.renderer = self.renderers[0]
        self.rays = 0
        self.gpu_ms = 0.0

    def image_height(self, width: int) -> int:
        return self.scene.image_height(width)

    def render_band(self, width: int, height: int, lo: int, hi: int, s0: int, n: int, max_bounces: int, seed: int):
        """f64 radiance sums of rows [lo, hi) over samples [s0, s0 + n): a device tensor of (hi-lo)*width*3."""
        torch = self.torch
        accums = [torch.zeros((hi - lo) * width * 3, dtype=torch.float64, device=self.device) for _ in self.renderers]
        if n > 0:
            # torch's stream -> the library's streams.  Only the stream that zeroed the buffers: a device-wide
            # synchronize would also wait for the band before, whose reduce is still running on RCCL's stream -
            # the overlap of a band's reduce with the next band's tracing would be gone.
            torch.cuda.current_stream(self.device).synchronize()
            parts = parallel.render_concurrent(self.renderers, accums, width, height, lo, hi, s0, n, max_bounces, seed)
            self.rays += sum(st.rays for st in parts)
            self.gpu_ms += max(st.ms_total for st in parts)
        return accums[0]

    def quantize(self, accum, n_pixels: int, spp: int) -> np.ndarray:
        torch = self.torch
        rgb8 = torch.empty(n_pixels * 3, dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        self.renderer.quantize_device(accum.data_ptr(), n_pixels, spp, rgb8.data_ptr())
        return rgb8.cpu().numpy()

    def tensor(self, values, dtype):
        return self.torch.tensor(values, dtype=dtype, device=self.device)


def plan_bands(height: int, present: np.ndarray, band_rows: int) -> List[Tuple[int, int]]:
    """Runs of rows missing from the part file, from the top of the picture (j = height - 1) down,
    cut into bands of at most band_rows rows: [(lo, hi)] with rows [lo, hi), in render order."""
    bands = []
    hi = height
    while hi > 0:
        if present[hi - 1]:
            hi -= 1
            continue
        lo = hi - 1
        while lo > 0 and not present[lo - 1] and hi - (lo - 1) <= band_rows:
            lo -= 1
        bands.append((lo, hi))
        hi = lo
    return bands


def default_band_rows(width: int, spp: int, world: int) -> int:
    """Rows per band so that a rank has about 2^24 paths of a band in flight (bin/raytracer's rule, per rank)."""
    per_row = width * max(1, -(-max(spp, 1) // world))
    return max(1, (1 << 24) // per_row)


class _Monitor(threading.Thread):
    """monitor_progress (src/main.rs:191-246) on rank 0: progress bar + one .PART record per finished row."""

    def __init__(self, rt, output_file: str, width: int, height: int, quiet: bool, no_part_file: bool, glyphs: str,
                 bar_len: int, bar_samples: int):
        super().__init__(daemon=True)
        self.rows: "queue.Queue" = queue.Queue()
        self.error: Optional[BaseException] = None
        self.quiet = quiet
        self.height = height
        self.part = None if no_part_file else rt.PartFile(output_file, width, height)
        self.bar = None if quiet else rt.ProgressBar(bar_len, "Rendering", glyphs, bar_samples, height)

    def run(self):
        complete = True
        try:
            for _ in range(self.height):
                msg = self.rows.get()
                if msg is None:  # the renderer stopped early: keep the part file for --recover-from
                    complete = False
                    break
                if self.bar:
                    self.bar.update()
                if self.part:
                    try:
                        self.part.write(msg[0], msg[1])
                    except Exception as e:  # src/main.rs:226-231
                        sys.stderr.write(f"\rError writing to part file: {e}\nPart file may be corrupted\n")
            if self.part:
                self.part.finish(remove=complete)
            if self.bar:
                self.bar.clear()
        except BaseException as e:  # noqa: BLE001 - reported by the render thread
            self.error = e


def render_image(engine, rt, *, width: int, spp: int, max_bounces: int, seed: int, output_file: Optional[str],
                 rank: int = 0, world: int = 1, dist=None, recover_from: Optional[str] = None, recover_corrupt: bool = False,
                 no_part_file: bool = False, quiet: bool = True, glyphs: str = " -=≡", bar_len: int = 60,
                 bar_samples: int = 100, band_rows: int = 0) -> Optional[np.ndarray]:
    """Renders the engine's scene on `world` ranks; returns the [H, W, 3] uint8 picture (top row first) on
    rank 0 and None elsewhere.  `rt` is the module that provides PartFile / read_part_file / ProgressBar /
    save_image (raytracer_amd).  Raises on every rank if rank 0 cannot read the part file."""
    if width < 2:
        raise ValueError("width must be at least 2")
    height = engine.image_height(width)
    # Rank 0 reads the part file; every rank needs to know which rows it covers to plan the same bands.
    base = None
    present = np.zeros(height, dtype=np.uint8)
    status = 0
    err = None
    if rank == 0 and recover_from:
        try:
            base, ok = rt.read_part_file(recover_from, height, width, recover_corrupt)
            present = ok.astype(np.uint8)
        except Exception as e:  # noqa: BLE001
            status, err = 1, e
    if dist is not None and world > 1:
        import torch

        flags = engine.tensor([status] + present.tolist(), torch.int32)
        dist.broadcast(flags, src=0)
        flags = flags.cpu().numpy()
        status, present = int(flags[0]), flags[1:].astype(np.uint8)
    if status:
        raise err if err is not None else RuntimeError("rank 0 could not read the part file")

    bands = plan_bands(height, present, band_rows if band_rows > 0 else default_band_rows(width, spp, world))
    # Fewer samples than ranks: the ROWS of a band are split instead (parallel.split_rows) - every rank traces all the
    # samples of its rows into a zeroed band buffer, so the same reduce assembles the band and no sum is reordered.
    tile_split = world > 1 and spp < world
    s0, n = (0, spp) if tile_split else parallel.split_samples(spp, world, rank)

    pixels = monitor = None
    if rank == 0:
        pixels = np.zeros((height, width, 3), dtype=np.uint8)
        monitor = _Monitor(rt, output_file or "image", width, height, quiet, no_part_file or output_file is None, glyphs,
                           bar_len, bar_samples)
        monitor.start()

    def emit(j: int, row: np.ndarray) -> None:
        pixels[height - 1 - j] = row
        monitor.rows.put((j, row))

    def agree(error: Optional[BaseException]) -> None:
        """Every rank learns whether any rank failed BEFORE the next collective: a rank that left the loop alone would
        leave its peers waiting in the reduce until the RCCL timeout."""
        failed = 1 if error is not None else 0
        if dist is not None and world > 1:
            import torch

            flag = engine.tensor([failed], torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = int(flag.cpu().numpy()[0])
        if error is not None:
            raise error
        if failed:
            raise RuntimeError("another rank failed while rendering; see its message")

    def trace(lo: int, band_hi: int):
        if not tile_split:
            return engine.render_band(width, height, lo, band_hi, s0, n, max_bounces, seed)
        r0, r1 = parallel.split_rows(band_hi - lo, world, rank)
        accum = engine.render_band(width, height, lo, band_hi, 0, 0, max_bounces, seed)  # zeros
        if r1 > r0:
            part = engine.render_band(width, height, lo + r0, lo + r1, 0, spp, max_bounces, seed)
            accum[r0 * width * 3:r1 * width * 3] = part
        return accum

    # A band's reduce runs while the next band is traced (asynchronous collective, one band in flight); the rows are
    # handed to the monitor in descending j all the same: what follows a band waits until that band is finished.
    todo: List = []  # what rank 0 still has to emit, in order: ("rows", [j...]) or ("band", lo, hi, accum, work)

    def flush(keep_last_band: bool) -> None:
        while todo:
            item = todo[0]
            if item[0] == "band":
                if keep_last_band and len([t for t in todo if t[0] == "band"]) == 1:
                    return
                _, lo, band_hi, accum, work = item
                if work is not None:
                    work.wait()
                if rank == 0:
                    rgb = engine.quantize(accum, (band_hi - lo) * width, spp).reshape(band_hi - lo, width, 3)
                    for j in range(band_hi - 1, lo - 1, -1):
                        emit(j, rgb[j - lo])
            elif rank == 0:
                for j in item[1]:
                    emit(j, base[j])
            todo.pop(0)

    try:
        hi = height
        for lo, band_hi in bands:
            todo.append(("rows", list(range(hi - 1, band_hi - 1, -1))))  # recovered rows above this band
            error = accum = None
            try:
                accum = trace(lo, band_hi)
            except BaseException as e:  # noqa: BLE001 - raised again by agree(), on every rank
                error = e
            agree(error)
            todo.append(("band", lo, band_hi, accum, parallel.reduce_to_root_async(accum, dist)))
            flush(keep_last_band=True)
            hi = lo
        todo.append(("rows", list(range(hi - 1, -1, -1))))
        flush(keep_last_band=False)
    except BaseException:
        if monitor is not None:
            monitor.rows.put(None)
            monitor.join()
        raise
    if rank != 0:
        return None
    monitor.join()
    if monitor.error is not None:
        raise monitor.error
    if output_file is not None:
        rt.save_image(output_file, pixels)
    return pixels


def _terminal_width() -> int:  # terminal_size(), src/main.rs:209-213
    for fd in (1, 2, 0):
        try:
            if os.isatty(fd):
                cols = os.get_terminal_size(fd).columns
                if cols > 0:
                    return min(cols, 100)
        except OSError:
            pass
    return 60


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="raytracer_amd.render", description="FBemf/raytracer's command line over the GPUs of one node")
    ap.add_argument("input_file")
    ap.add_argument("output_file")
    ap.add_argument("-w", "--width", type=int, default=600)
    ap.add_argument("-s", "--ray-samples", type=int, default=100)
    ap.add_argument("-m", "--max-bounces", type=int, default=50)
    ap.add_argument("-q", "--quiet", action="store_true")
    ap.add_argument("--ascii-symbols-only", action="store_true")
    ap.add_argument("-r", "--recover-from")
    ap.add_argument("--recover-corrupt", action="store_true")
    ap.add_argument("--progress-bar-len", type=int)
    ap.add_argument("--progress-bar-samples", type=int, default=100)
    ap.add_argument("--no-part-file", action="store_true")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--overlap", type=int, default=1,
                    help="concurrent batches per GPU (2: about 12 %% faster; the picture may differ from --overlap 1 in the last bit of a sum)")
    args = ap.parse_args(argv)

    if args.overlap > 1:  # two streams share the CUs: 8 persistent blocks per CU each measure best
        os.environ.setdefault("RT_GRID_MULT", "8")

    import torch

    import raytracer_amd as rt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.stderr.write("Error: no HIP device (the renderer has no CPU path)\n")
        return 1
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    start = time.monotonic()
    rc = 0
    try:
        engine = GpuEngine(args.input_file, local_rank, args.precision, overlap=args.overlap)
        seed = args.seed
        if seed is None:  # one seed for the whole job, drawn on rank 0
            seed = int.from_bytes(os.urandom(8), "little") >> 1
            if dist is not None:
                t = engine.tensor([seed], torch.int64)
                dist.broadcast(t, src=0)
                seed = int(t.item())
        render_image(engine, rt, width=args.width, spp=args.ray_samples, max_bounces=args.max_bounces, seed=seed,
                     output_file=args.output_file, rank=rank, world=world, dist=dist, recover_from=args.recover_from,
                     recover_corrupt=args.recover_corrupt, no_part_file=args.no_part_file, quiet=args.quiet,
                     glyphs=" .:" if args.ascii_symbols_only else " -=≡",
                     bar_len=args.progress_bar_len if args.progress_bar_len else _terminal_width(),
                     bar_samples=args.progress_bar_samples, band_rows=args.band_rows)
        if args.stats:
            t = engine.tensor([float(engine.rays), engine.gpu_ms], torch.float64)
            if dist is not None:
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                wall = time.monotonic() - start
                sys.stderr.write(f"seed {seed}: {int(t[0].item())} rays on {world} GPU(s), {t[1].item() / world:.2f} ms on each GPU (mean), "
                                 f"{wall:.2f} s wall clock incl. scene load\n")
        if rank == 0 and not args.quiet:
            el = int(time.monotonic() - start)
            sys.stderr.write(f"Completed in {el // 60}:{el % 60:02d}\n")
    except Exception as e:  # noqa: BLE001 - anyhow's report of main's Err
        sys.stderr.write(f"Error: {e}\n")
        rc = 1
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
