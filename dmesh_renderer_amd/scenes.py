"""Synthetic scenes for parity tests and bench.py.

The reference ships no scenes, tests or benchmarks, so these are
the build's own workloads: "layered sheets" for the tri renderer and a Kuhn-split cube
lattice for the tet renderer.  Everything is generated on the CPU with a seeded torch
generator so the same arrays are produced in the container and on the GPU box.

Matrices are returned ROW-major ([B,4,4], as a DMesh caller would hold them); the
renderer modules transpose them (reference dmesh_renderer/__init__.py:219-220).
"""
from __future__ import annotations

import math
from typing import Dict, NamedTuple

import numpy as np
import torch as th


class Config(NamedTuple):
    name: str
    kind: str      # "tri" | "tet"
    layers: int    # tri: sheets L ; tet: unused
    n: int         # tri: lattice points per side ; tet: cubes per side m
    B: int
    H: int
    W: int


# BASELINE.json configs (C1..C5).  F/P per SURVEY 8(d).
CONFIGS: Dict[str, Config] = {
    "C1": Config("tri 2k tris 256x256", "tri", 4, 17, 1, 256, 256),
    "C2": Config("tri 100k tris 800x800", "tri", 8, 80, 1, 800, 800),
    "C3": Config("tet 50k faces 800x800", "tet", 0, 16, 1, 800, 800),
    "C4": Config("tri 500k tris 1920x1080", "tri", 16, 126, 1, 1080, 1920),
    "C5": Config("tri 2M tris 4096x4096 B4", "tri", 16, 251, 4, 4096, 4096),
}


def perspective(fovy_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    f = 1.0 / math.tan(math.radians(fovy_deg) / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = (far + near) / (near - far)
    m[2, 3] = 2.0 * far * near / (near - far)
    m[3, 2] = -1.0
    return m


def look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)) -> np.ndarray:
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    side = np.cross(fwd, np.asarray(up, dtype=np.float64))
    side /= np.linalg.norm(side)
    upv = np.cross(side, fwd)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = side, upv, -fwd
    m[:3, 3] = -m[:3, :3] @ eye
    return m


def cameras(B: int, H: int, W: int, radius: float = 3.0, elev: float = 0.0):
    """B cameras on a circle of `radius` around the origin, azimuth 360*b/B."""
    mv = np.zeros((B, 4, 4), dtype=np.float64)
    proj = np.zeros((B, 4, 4), dtype=np.float64)
    for b in range(B):
        az = 2.0 * math.pi * b / B
        eye = (radius * math.sin(az) * math.cos(elev), radius * math.sin(elev),
               radius * math.cos(az) * math.cos(elev))
        mv[b] = look_at(eye)
        proj[b] = perspective(60.0, W / H, 0.1, 10.0)
    return mv, proj


def _ndc_z(verts: np.ndarray, mv: np.ndarray, proj: np.ndarray) -> np.ndarray:
    vh = np.concatenate([verts.astype(np.float64), np.ones((verts.shape[0], 1))], axis=1)
    out = np.zeros((mv.shape[0], verts.shape[0]), dtype=np.float64)
    for b in range(mv.shape[0]):
        clip = vh @ (proj[b] @ mv[b]).T
        out[b] = clip[:, 2] / clip[:, 3]
    return out


def layered_sheets(L: int, n: int, B: int, H: int, W: int, seed: int = 0,
                   opacity=(0.1, 0.5)) -> Dict[str, th.Tensor]:
    """L semi-transparent n x n jittered lattices stacked in z in [-0.8, 0.8]."""
    g = th.Generator().manual_seed(seed)
    cell = 2.0 / (n - 1)
    lin = th.linspace(-1.0, 1.0, n, dtype=th.float64)
    yy, xx = th.meshgrid(lin, lin, indexing="ij")
    zs = th.linspace(-0.8, 0.8, L, dtype=th.float64) if L > 1 else th.zeros(1, dtype=th.float64)
    verts = th.zeros(L, n, n, 3, dtype=th.float64)
    jit = (th.rand(L, n, n, 2, generator=g, dtype=th.float64) * 0.6 - 0.3) * cell
    verts[..., 0] = xx[None] + jit[..., 0]
    verts[..., 1] = yy[None] + jit[..., 1]
    verts[..., 2] = zs[:, None, None]
    verts = verts.reshape(-1, 3)
    ii, jj = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    v00 = (ii * n + jj).reshape(-1)
    v01, v10, v11 = v00 + 1, v00 + n, v00 + n + 1
    quad = np.concatenate([np.stack([v00, v01, v11], 1), np.stack([v00, v11, v10], 1)], 0)
    faces = np.concatenate([quad + l * n * n for l in range(L)], 0).astype(np.int32)
    P, F = verts.shape[0], faces.shape[0]
    mv, proj = cameras(B, H, W)
    vnp = verts.numpy()
    return {
        "verts": verts.to(th.float32),
        "faces": th.from_numpy(faces),
        "verts_color": th.rand(P, 3, generator=g, dtype=th.float64).to(th.float32),
        "faces_opacity": (th.rand(F, generator=g, dtype=th.float64) * (opacity[1] - opacity[0]) + opacity[0]).to(th.float32),
        "faces_intense": (th.rand(B, F, generator=g, dtype=th.float64) * 0.5 + 0.5).to(th.float32),
        "verts_depth": th.from_numpy(_ndc_z(vnp, mv, proj)).to(th.float32),
        "mv_mats": th.from_numpy(mv).to(th.float32),
        "proj_mats": th.from_numpy(proj).to(th.float32),
        "bg": th.zeros(3, dtype=th.float32),
    }


_KUHN_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


def kuhn_tets(m: int, B: int, H: int, W: int, seed: int = 0, opacity=(0.02, 0.3),
              jitter: float = 0.25) -> Dict[str, th.Tensor]:
    """Kuhn 6-tet split of an m^3 cube lattice over [-1,1]^3, interior vertices jittered.

    Returns tets [T,4], faces [F,3], face_tets [F,2] (-1 = boundary), tet_faces [T,4]."""
    g = th.Generator().manual_seed(seed)
    n = m + 1
    lin = np.linspace(-1.0, 1.0, n)
    zz, yy, xx = np.meshgrid(lin, lin, lin, indexing="ij")
    verts = np.stack([xx, yy, zz], -1).reshape(-1, 3)
    cell = 2.0 / m
    jit = (th.rand(n, n, n, 3, generator=g, dtype=th.float64).numpy() * 2.0 - 1.0) * jitter * cell
    interior = np.zeros((n, n, n), dtype=bool)
    interior[1:-1, 1:-1, 1:-1] = True
    verts = verts + (jit * interior[..., None]).reshape(-1, 3)

    def vid(i, j, k):  # x index i, y index j, z index k
        return (k * n + j) * n + i

    ci, cj, ck = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    ci, cj, ck = ci.reshape(-1), cj.reshape(-1), ck.reshape(-1)
    tets = []
    for perm in _KUHN_PERMS:
        cur = np.stack([ci, cj, ck], 1)
        ids = [vid(cur[:, 0], cur[:, 1], cur[:, 2])]
        for ax in perm:
            cur = cur.copy()
            cur[:, ax] += 1
            ids.append(vid(cur[:, 0], cur[:, 1], cur[:, 2]))
        tets.append(np.stack(ids, 1))
    tets = np.concatenate(tets, 0).astype(np.int64)
    T = tets.shape[0]
    # faces = unique sorted vertex triples
    tri_idx = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
    tri = np.sort(tets[:, tri_idx].reshape(-1, 3), axis=1)
    key = (tri[:, 0] * (verts.shape[0] + 1) + tri[:, 1]) * (verts.shape[0] + 1) + tri[:, 2]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    F = uniq.shape[0]
    faces = tri[first].astype(np.int32)
    tet_faces = inv.reshape(T, 4).astype(np.int32)
    face_tets = np.full((F, 2), -1, dtype=np.int32)
    owner = np.repeat(np.arange(T), 4)
    srt = np.argsort(inv, kind="stable")
    inv_s, own_s = inv[srt], owner[srt]
    start = np.searchsorted(inv_s, np.arange(F), side="left")
    cnt = np.searchsorted(inv_s, np.arange(F), side="right") - start
    face_tets[:, 0] = own_s[start]
    two = cnt > 1
    face_tets[two, 1] = own_s[start[two] + 1]
    P = verts.shape[0]
    mv, proj = cameras(B, H, W)
    return {
        "verts": th.from_numpy(verts).to(th.float32),
        "faces": th.from_numpy(faces),
        "tets": th.from_numpy(tets.astype(np.int32)),
        "face_tets": th.from_numpy(face_tets),
        "tet_faces": th.from_numpy(tet_faces),
        "verts_color": th.rand(P, 3, generator=g, dtype=th.float64).to(th.float32),
        "faces_opacity": (th.rand(F, generator=g, dtype=th.float64) * (opacity[1] - opacity[0]) + opacity[0]).to(th.float32),
        "faces_intense": (th.rand(B, F, generator=g, dtype=th.float64) * 0.5 + 0.5).to(th.float32),
        "verts_depth": th.from_numpy(_ndc_z(verts, mv, proj)).to(th.float32),
        "mv_mats": th.from_numpy(mv).to(th.float32),
        "proj_mats": th.from_numpy(proj).to(th.float32),
        "bg": th.zeros(3, dtype=th.float32),
    }


def make(config: str, seed: int = 0, **over) -> Dict[str, th.Tensor]:
    c = CONFIGS[config]
    if c.kind == "tri":
        return layered_sheets(over.get("layers", c.layers), over.get("n", c.n), over.get("B", c.B),
                              over.get("H", c.H), over.get("W", c.W), seed=seed,
                              opacity=over.get("opacity", (0.1, 0.5)))
    return kuhn_tets(over.get("n", c.n), over.get("B", c.B), over.get("H", c.H), over.get("W", c.W),
                     seed=seed, opacity=over.get("opacity", (0.02, 0.3)))


def c_args(d: Dict[str, th.Tensor], device=None, tet: bool = False):
    """Scene dict (row-major matrices) -> the leading tensor arguments of `_C.render_tris` /
    `_C.render_tets`: transposed mv/proj and their inverses, as the reference wrapper passes them
    (dmesh_renderer/__init__.py:62-63,219-220)."""
    mv_t = d["mv_mats"].transpose(1, 2)
    proj_t = d["proj_mats"].transpose(1, 2)
    inv_mv, inv_proj = th.inverse(mv_t), th.inverse(proj_t)
    args = [d["bg"], d["verts"], d["faces"], d["verts_color"], d["faces_opacity"], mv_t, proj_t, inv_mv, inv_proj,
            d["verts_depth"], d["faces_intense"]]
    if tet:
        args += [d["tets"], d["face_tets"], d["tet_faces"]]
    if device is not None:
        args = [a.to(device) for a in args]
    return args


def upstream_grads(B: int, H: int, W: int, seed: int = 1):
    """dL/dcolor [B,3,H,W], dL/ddepth [B,1,H,W] ~ N(0,1), seeded."""
    g = th.Generator().manual_seed(seed)
    return th.randn(B, 3, H, W, generator=g), th.randn(B, 1, H, W, generator=g)


def rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    """NORMALISED error max-abs(got - ref) / max(1, max-abs(ref)): SURVEY 8(d)'s definition of the metric's gradient
    error (a whole-tensor figure: entries much smaller than the largest one are only checked in absolute terms; see
    max_abs_err for the plain difference and elementwise_close for a per-entry check)."""
    if ref.size == 0:
        return 0.0
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(1.0, float(np.abs(ref).max())))


def max_abs_err(got: np.ndarray, ref: np.ndarray) -> float:
    """max |got - ref|, not normalised."""
    if ref.size == 0:
        return 0.0
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())


# Two float evaluations of the same tri gradients that differ in summation order only (bands vs full frame, two launches:
# which records share a lane is decided by claim order) agree to this, normalised like rel_err; dL_dverts sets it (see
# elementwise_close), the other four tensors repeat to ~1e-7.
SUM_ORDER_TOL = 5e-5


def sum_order_tol(name: str) -> float:
    """The tolerance between two evaluations that differ in summation order only, per gradient tensor:
    SUM_ORDER_TOL is dL_dverts' (the ray-moment sums), the other four tensors repeat to ~1e-7 and keep 1e-5."""
    return SUM_ORDER_TOL if name == "verts" else 1e-5


def elementwise_close(got: np.ndarray, ref: np.ndarray, rtol: float = 1e-3, atol_scale: float = 2e-5) -> bool:
    """Per entry: |got - ref| <= atol + rtol * |ref| with atol = atol_scale * max(1, max-abs(ref)).  A gradient entry is
    a sum of many signed per-pixel terms (float atomics in the reference, table sums here), so its absolute error
    scales with the tensor, not with the entry; the rtol term makes every entry that stands out of that noise floor
    agree to 0.1 %.  atol_scale = 2e-5 is a fifth of the 1e-4 bar; what sets the floor is the reference formula's own
    float rounding in dL_dverts (cross(T, d) of two nearly parallel vectors): the oracle in float is 1.1e-4 (C4) / 1.7e-3 (C5)
    of the largest entry away from the same formula evaluated in double, the library's evaluation order 1.2e-5 / 1.3e-5 away
    from the oracle (tests/tools/grad_noise.py, profiles/r02/grad_noise.txt)."""
    if ref.size == 0:
        return True
    g, r = got.astype(np.float64), ref.astype(np.float64)
    atol = atol_scale * max(1.0, float(np.abs(r).max()))
    return bool((np.abs(g - r) <= atol + rtol * np.abs(r)).all())
