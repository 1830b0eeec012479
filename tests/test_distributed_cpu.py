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
