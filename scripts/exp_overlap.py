"""Experiment: one image's samples as K concurrent batches (K contexts = K streams on one GPU, one host thread
each), so that one batch's latency-bound late bounces and tail launch run under another batch's early bounces.
usage: python scripts/exp_overlap.py <scene> [spp] [steps]     (prints ms per image for K = 1, 2, 3, 4)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import raytracer_amd as rt  # noqa: E402
from raytracer_amd import parallel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "knight"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scene = rt.Scene.load(os.path.join(ROOT, "examples", name + ".json5"), ROOT)
W = 600
H = scene.image_height(W)
dev = torch.device("cuda", 0)
for K in (1, 2, 3, 4):
    rens = []
    for _ in range(K):
        r = rt.Renderer(device=0, max_paths=(1 << 26) // K)
        r.upload(scene)
        rens.append(r)
    accs = [torch.zeros(W * H * 3, dtype=torch.float64, device=dev) for _ in range(K)]
    rays = [0] * K

    def work(k):
        s0, n = parallel.split_samples(spp, K, k)
        st = rens[k].render_accum_device(W, H, 0, H, s0, n, 50, 0, accs[k].data_ptr())
        rays[k] = st.rays

    def image():
        for a in accs:
            a.zero_()
        torch.cuda.synchronize()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        total = accs[0]
        for a in accs[1:]:
            total = total + a
        torch.cuda.synchronize()
        return total

    ref = image()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = image()
    dt = (time.perf_counter() - t0) / steps
    if K == 1:
        base = out.clone()
    err = float((out - base).abs().max())
    print(f"{name} spp {spp} K={K}: {dt * 1e3:.2f} ms per image, {sum(rays) / dt / 1e6:.1f} Mrays/s, max |sum - sum(K=1)| = {err:.3e}", flush=True)
    for r in rens:
        r.close()
