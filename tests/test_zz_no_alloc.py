"""rt_render_accum_device_noalloc: the render entry point that allocates nothing (work buffers in the library's static
device array) must return exactly what rt_render_accum_device returns, call after call."""
import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


def _both(rt, r, torch, W, H, j0, j1, s0, ns, bounces, seed):
    n = (j1 - j0) * W * 3
    want = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    got = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    r.render_accum_device(W, H, j0, j1, s0, ns, bounces, seed, want.data_ptr(), want_stats=False)
    r.render_accum_device_noalloc(W, H, j0, j1, s0, ns, bounces, seed, got.data_ptr())
    torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy()


@pytest.mark.parametrize("scene_name", ["cornell", "knight"])
def test_noalloc_equals_the_allocating_entry_point_twice_in_a_row(rt, scene_name):
    import torch

    scene = rt.Scene.load(scene_path(scene_name), ROOT)
    r = rt.Renderer(device=0)
    r.upload(scene)
    # first call: 64 x 64 pixels x 40 samples = 163840 paths, more than the static area's 131072 -> two batches
    W = 64
    H = scene.image_height(W)
    want, got = _both(rt, r, torch, W, H, 0, H, 3, 40, 12, seed=5)
    assert want.any()
    assert np.array_equal(want, got)  # same samples added in the same order: bit-identical
    # second call, smaller input (a band of a smaller picture, fewer samples): the array still holds the first call's data
    W2 = 32
    H2 = scene.image_height(W2)
    want2, got2 = _both(rt, r, torch, W2, H2, 5, 20, 0, 7, 50, seed=6)
    assert want2.any()
    assert np.array_equal(want2, got2)
    # and it adds to what accum holds, like the allocating call
    acc = torch.ones((20 - 5) * W2 * 3, dtype=torch.float64, device="cuda:0")
    ref = torch.ones((20 - 5) * W2 * 3, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    r.render_accum_device_noalloc(W2, H2, 5, 20, 0, 7, 50, 6, acc.data_ptr())
    r.render_accum_device(W2, H2, 5, 20, 0, 7, 50, 6, ref.data_ptr(), want_stats=False)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), ref.cpu().numpy()) and not np.array_equal(acc.cpu().numpy(), want2)


def test_noalloc_rejects_bands_beyond_the_static_area(rt):
    import torch

    scene = rt.Scene.load(scene_path("cornell"), ROOT)
    r = rt.Renderer(device=0)
    r.upload(scene)
    W = 600
    H = scene.image_height(W)
    rows = rt.NOALLOC_MAX_PATHS // W + 1
    acc = torch.zeros(rows * W * 3, dtype=torch.float64, device="cuda:0")
    with pytest.raises(rt.RtError) as e:
        r.render_accum_device_noalloc(W, H, 0, rows, 0, 1, 4, 0, acc.data_ptr())
    assert e.value.code == -1 and "131072" in e.value.message
    # the largest band that fits still renders
    r.render_accum_device_noalloc(W, H, 0, rows - 1, 0, 1, 4, 0, acc.data_ptr())
    torch.cuda.synchronize()
    assert float(acc.sum()) > 0.0


def test_noalloc_in_f32_mode(rt):
    import torch

    scene = rt.Scene.load(scene_path("cornell"), ROOT)
    r = rt.Renderer(device=0, precision=rt.PRECISION_F32)
    r.upload(scene)
    W = 48
    H = scene.image_height(W)
    want, got = _both(rt, r, torch, W, H, 0, H, 0, 9, 8, seed=1)
    assert want.any() and np.array_equal(want, got)


def test_two_contexts_on_one_gpu_do_not_share_the_work_area_at_once(rt):
    # Every context on a GPU carves its buffers from the same static array: calls that overlap are refused (RT_ERR_BUSY = -7)
    # instead of corrupting each other's queues.  Two contexts, two threads, several rounds: every call either returns the
    # allocating entry point's sums bit for bit or is refused; calls made one after the other all succeed.
    import threading

    import torch

    sc = rt.Scene.load(scene_path("cornell"), ROOT)
    W = 128
    H = sc.image_height(W)
    ctxs = [rt.Renderer(0), rt.Renderer(0)]
    for r in ctxs:
        r.upload(sc)
    want = torch.zeros(W * H * 3, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    ctxs[0].render_accum_device(W, H, 0, H, 0, 24, 50, 3, want.data_ptr(), want_stats=False)
    outcomes = [[], []]

    def worker(k):
        for _ in range(6):
            acc = torch.zeros(W * H * 3, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            try:
                ctxs[k].render_accum_device_noalloc(W, H, 0, H, 0, 24, 50, 3, acc.data_ptr())
                outcomes[k].append(bool(torch.equal(acc, want)))
            except rt.RtError as e:
                assert e.code == -7 and "in use" in e.message
                outcomes[k].append(None)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    done = [o for o in outcomes[0] + outcomes[1] if o is not None]
    assert done and all(done), outcomes          # whatever ran, ran alone and is right
    for r in ctxs:                               # and one after the other, nothing is refused
        acc = torch.zeros(W * H * 3, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        r.render_accum_device_noalloc(W, H, 0, H, 0, 24, 50, 3, acc.data_ptr())
        assert torch.equal(acc, want)
