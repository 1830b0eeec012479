"""world_size-2 test of the multi-GPU path's host logic on CPU (gloo): sample ranges are split
across ranks, per-rank radiance sums are reduced to rank 0, and the result equals the
single-process render.  The per-rank renderer here is the CPU oracle (this is a test of the
partition + reduce, not of the kernels; bench.py runs the same helpers with the GPU renderer
and the nccl/RCCL backend)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_spp, out_path):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from oracle import binding as orc
    from raytracer_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = orc.OracleScene.load(os.path.join(ROOT, "examples", "cornell.json5"), ROOT)
    W = H = 20

    def render_range(s0, n):
        if n == 0:
            return torch.zeros(H * W * 3, dtype=torch.float64)
        _, sums, _ = scene.render(W, H, 0, H, n, 10, seed=11, s0=s0)
        return torch.from_numpy(sums.reshape(-1).copy())

    accum, (s0, n) = parallel.render_distributed(render_range, total_spp, rank, world, dist)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_spp", [7, 1])
def test_two_rank_sample_split_equals_single_process(tmp_path, total_spp, orc):
    import torch.multiprocessing as mp

    out = str(tmp_path / "accum.npy")
    mp.spawn(_worker, args=(2, _free_port(), total_spp, out), nprocs=2, join=True)
    got = np.load(out)
    scene = orc.OracleScene.load(os.path.join(ROOT, "examples", "cornell.json5"), ROOT)
    _, want, _ = scene.render(20, 20, 0, 20, total_spp, 10, seed=11)
    assert np.allclose(got, want.reshape(-1), rtol=1e-13, atol=0)


class _OracleEngine:
    """The engine interface of raytracer_amd.render.render_image over the CPU oracle (test only)."""

    def __init__(self, orc, bounces):
        self.scene = orc.OracleScene.load(os.path.join(ROOT, "examples", "cornell.json5"), ROOT)
        self.bounces = bounces

    def image_height(self, width):
        return self.scene.image_height(width)

    def render_band(self, width, height, lo, hi, s0, n, max_bounces, seed):
        import torch

        if n == 0:
            return torch.zeros((hi - lo) * width * 3, dtype=torch.float64)
        _, sums, _ = self.scene.render(width, height, lo, hi, n, max_bounces, seed=seed, s0=s0)
        return torch.from_numpy(sums.reshape(-1).copy())

    def quantize(self, accum, n_pixels, spp):  # src/main.rs:138-142, 172-177
        c = np.sqrt(accum.numpy() / spp)
        return np.floor(255.0 * np.clip(np.abs(c), 0.0, 0.999)).astype(np.uint8)

    def tensor(self, values, dtype):
        import torch

        return torch.tensor(values, dtype=dtype)


def _image_worker(rank, world, port, workdir, spp, recover):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    from oracle import binding as orc
    from raytracer_amd import render

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(workdir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pixels = render.render_image(_OracleEngine(orc, 6), rt, width=24, spp=spp, max_bounces=6, seed=3, output_file="node.ppm",
                                     rank=rank, world=world, dist=dist, recover_from=recover, band_rows=5, quiet=True)
        assert (pixels is not None) == (rank == 0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("spp", [5, 1])
def test_two_rank_image_with_bands_and_part_file(tmp_path, spp, orc):
    """raytracer_amd.render.render_image on two gloo ranks (oracle engine): bands from the top down, rows
    recovered from a part file kept verbatim and not rendered, one reduce per band, the file written by
    rank 0 equals the single-process picture, and no part file is left behind."""
    import torch.multiprocessing as mp

    import raytracer_amd as rt

    W = H = 24
    scene = orc.OracleScene.load(os.path.join(ROOT, "examples", "cornell.json5"), ROOT)
    want, _, _ = scene.render(W, H, 0, H, spp, 6, seed=3)  # rows indexed by j (bottom-up)
    kept = [23, 22, 14, 13, 12, 11, 10, 9, 0]
    marker = np.full((W, 3), 7, dtype=np.uint8)
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        pf = rt.PartFile("node.ppm", W, H)
        for j in kept:
            pf.write(j, marker)
        pf.finish(remove=False)
    finally:
        os.chdir(old)
    mp.spawn(_image_worker, args=(2, _free_port(), str(tmp_path), spp, pf.path), nprocs=2, join=True)
    raw = open(tmp_path / "node.ppm", "rb").read()
    head = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(head)
    got = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(H, W, 3)
    for j in range(H):
        row = got[H - 1 - j]
        if j in kept:
            assert np.array_equal(row, marker), f"recovered row j={j}"
        else:  # two partial sums instead of one running sum: at most the last bit of a sum, so at most one step of a byte
            assert np.abs(row.astype(int) - want[j].astype(int)).max() <= 1 and (row != want[j]).sum() <= 1, f"row j={j}"
    assert sorted(os.listdir(tmp_path)) == ["node.ppm", "node.ppm.0.PART"]


def test_plan_bands():
    from raytracer_amd import render

    present = np.zeros(10, dtype=np.uint8)
    assert render.plan_bands(10, present, 4) == [(6, 10), (2, 6), (0, 2)]
    present[[9, 5, 4, 0]] = 1
    assert render.plan_bands(10, present, 3) == [(6, 9), (1, 4)]
    assert render.plan_bands(10, np.ones(10, dtype=np.uint8), 3) == []
    assert render.default_band_rows(600, 100, 1) == (1 << 24) // 60000
    assert render.default_band_rows(600, 100, 8) == (1 << 24) // (600 * 13)
    assert render.default_band_rows(3840, 10 ** 9, 1) == 1


class _FailingEngine(_OracleEngine):
    """Fails on one rank in the second band it is asked for."""

    def __init__(self, orc, bounces, fail):
        super().__init__(orc, bounces)
        self.fail = fail
        self.calls = 0

    def render_band(self, *args):
        self.calls += 1
        if self.fail and self.calls == 2:
            raise RuntimeError("injected failure on this rank")
        return super().render_band(*args)


def _failing_worker(rank, world, port, workdir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    from oracle import binding as orc
    from raytracer_amd import render

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(workdir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with pytest.raises(RuntimeError) as e:
            render.render_image(_FailingEngine(orc, 4, fail=(rank == 1)), rt, width=16, spp=4, max_bounces=4, seed=1,
                                output_file="fail.ppm", rank=rank, world=world, dist=dist, band_rows=4, quiet=True)
        assert ("injected failure" in str(e.value)) == (rank == 1)
        assert ("another rank failed" in str(e.value)) == (rank == 0)
        dist.barrier()  # both ranks are still in step: nobody is stuck in a reduce
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_stops_every_rank_before_the_next_collective(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert not os.path.exists(tmp_path / "fail.ppm")
    assert [f for f in os.listdir(tmp_path) if f.endswith(".PART")]  # kept for --recover-from
