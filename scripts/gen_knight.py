#!/usr/bin/env python3
"""Deterministic stand-in for examples/knight.obj.

The reference's examples/knight.json5:67 loads `examples/knight.obj`, but that asset is
not in the reference tree (/root/reference/.MISSING_LARGE_BLOBS:1).  This script writes a
procedural closed surface of the same role: one un-named object (no `o` line, so it is
selected by the default objectName ""), `v` / `vn` / `f v//vn` records, sitting at the place
the scene expects (after `translate [-11,-16,0.001]` it stands on the board near the origin,
z in [0, ~10], cf. examples/knight.json5:4-5,70-74).  Results obtained with it are labelled
"synthetic knight" everywhere.

The shape is a swept tube around a bent spine (a base, a neck leaning forward, a head) with a
sinusoidal displacement, so triangles come in many orientations and sizes.

usage: gen_knight.py OUT.obj [--triangles N]     (default about 100k triangles)
"""
import argparse
import math
import os

import numpy as np


def build(n_target: int):
    nt = max(8, int(round(math.sqrt(n_target / 2.0))))   # segments around
    nu = max(8, int(round(n_target / (2.0 * nt))))       # segments along the spine
    u = np.linspace(0.0, 1.0, nu + 1)
    th = np.linspace(0.0, 2.0 * math.pi, nt, endpoint=False)

    # Spine in the (x, z) plane: straight base, then a neck leaning towards -x, then a head.
    bend = np.clip((u - 0.35) / 0.65, 0.0, 1.0)
    sx = -2.6 * bend ** 2 + 1.1 * np.clip((u - 0.8) / 0.2, 0.0, 1.0) ** 2 * -1.0
    sz = 9.6 * u - 1.4 * bend ** 3
    spine = np.stack([sx, np.zeros_like(u), sz], axis=1)
    tangent = np.gradient(spine, u, axis=0)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    side = np.tile(np.array([0.0, 1.0, 0.0]), (nu + 1, 1))
    normal = np.cross(side, tangent)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)

    # Radius profile: wide foot, waist, chest, thin neck, head.
    def bump(c, w, a):
        return a * np.exp(-((u - c) / w) ** 2)
    radius = 1.1 + bump(0.0, 0.06, 1.7) + bump(0.12, 0.05, 0.5) + bump(0.45, 0.16, 0.9) + bump(0.86, 0.09, 0.75)
    radius *= np.clip(u / 0.01, 0.05, 1.0) ** 0.25
    radius *= np.clip((1.0 - u) / 0.04, 0.0, 1.0) ** 0.5 * 0.98 + 0.02

    U, TH = np.meshgrid(u, th, indexing="ij")
    R = radius[:, None] * (1.0 + 0.06 * np.sin(9.0 * TH + 14.0 * U) * np.sin(23.0 * U) + 0.03 * np.cos(5.0 * TH))
    # Flatten across y a little so it reads as a chess piece rather than a vase.
    P = (spine[:, None, :] + R[..., None] * (np.cos(TH)[..., None] * normal[:, None, :]
                                             + 0.8 * np.sin(TH)[..., None] * side[:, None, :]))
    # A muzzle: push the head forward along -x where theta points forward.
    head = np.clip((U - 0.78) / 0.12, 0.0, 1.0) * np.clip((1.0 - U) / 0.06, 0.0, 1.0)
    P[..., 0] -= 1.6 * head * np.clip(np.cos(TH - math.pi), 0.0, 1.0) ** 2

    # Per-vertex normals from the grid's partial derivatives (periodic in theta).
    dPu = np.gradient(P, axis=0)
    dPt = (np.roll(P, -1, axis=1) - np.roll(P, 1, axis=1)) * 0.5
    N = np.cross(dPt, dPu)
    ln = np.linalg.norm(N, axis=2, keepdims=True)
    N = np.where(ln > 1e-12, N / np.maximum(ln, 1e-12), np.array([0.0, 0.0, 1.0]))

    # Place it where the scene's translate expects it: base centre at (11, 16, 0).
    P = P + np.array([11.0, 16.0, 0.0])
    verts = P.reshape(-1, 3)
    norms = N.reshape(-1, 3)
    bottom = np.array([[spine[0, 0] + 11.0, 16.0, spine[0, 2]]])
    top = np.array([[spine[-1, 0] + 11.0, 16.0, spine[-1, 2]]])
    verts = np.concatenate([verts, bottom, top], axis=0)
    norms = np.concatenate([norms, [[0.0, 0.0, -1.0]], [tangent[-1]]], axis=0)

    idx = np.arange((nu + 1) * nt).reshape(nu + 1, nt)
    a = idx[:-1, :]
    b = np.roll(idx, -1, axis=1)[:-1, :]
    c = idx[1:, :]
    d = np.roll(idx, -1, axis=1)[1:, :]
    quads1 = np.stack([a, b, d], axis=-1).reshape(-1, 3)
    quads2 = np.stack([a, d, c], axis=-1).reshape(-1, 3)
    ib, it = (nu + 1) * nt, (nu + 1) * nt + 1
    cap_b = np.stack([np.full(nt, ib), np.roll(idx[0], -1), idx[0]], axis=-1)
    cap_t = np.stack([np.full(nt, it), idx[-1], np.roll(idx[-1], -1)], axis=-1)
    faces = np.concatenate([quads1, quads2, cap_b, cap_t], axis=0)
    return verts, norms, faces


def write_obj(path: str, verts, norms, faces) -> None:
    lines = ["# synthetic knight stand-in (scripts/gen_knight.py); the reference asset is missing"]
    lines += ["v %.6f %.6f %.6f" % tuple(v) for v in verts]
    lines += ["vn %.6f %.6f %.6f" % tuple(n) for n in norms]
    f1 = faces + 1
    lines += ["f %d//%d %d//%d %d//%d" % (f[0], f[0], f[1], f[1], f[2], f[2]) for f in f1]
    tmp = path + ".tmp"
    with open(tmp, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    os.replace(tmp, path)


def ensure(path: str, triangles: int = 100_000) -> str:
    """Writes the file unless it is already there; returns the path."""
    if not os.path.exists(path):
        v, n, f = build(triangles)
        write_obj(path, v, n, f)
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--triangles", type=int, default=100_000)
    args = ap.parse_args()
    v, n, f = build(args.triangles)
    write_obj(args.out, v, n, f)
    print(f"{args.out}: {len(v)} vertices, {len(f)} triangles")
