"""Shared helpers for the parity tests: run the same seeded scene through the CPU oracle
(oracle/, test infrastructure) and through the product's `_C` on the GPU."""
from __future__ import annotations

import numpy as np
import torch as th

from dmesh_renderer_amd import scenes

TRI_KEYS = ("verts", "faces", "verts_color", "faces_opacity")


def c_args(d: dict, device=None, tet: bool = False):
    """Module-convention scene dict -> the leading tensor arguments of _C.render_tris/_tets."""
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


def upstream_grads(B, H, W, seed=1):
    g = th.Generator().manual_seed(seed)
    return th.randn(B, 3, H, W, generator=g), th.randn(B, 1, H, W, generator=g)


def rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    """max-abs(got - ref) / max(1, max-abs(ref))   """
    if ref.size == 0:
        return 0.0
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(1.0, float(np.abs(ref).max())))


def tile_lists(ranges: np.ndarray, values: np.ndarray):
    r = ranges.reshape(-1, 2)
    return [values[a:b] for a, b in r]
