"""Oracle-backed stand-in for `_C` on CPU tensors (TEST INFRASTRUCTURE ONLY).

Same four functions and tuple arities as dmesh_renderer_amd._C / the reference ext.cpp:6-11, computed
by the CPU oracle.  Used (a) to run the REFERENCE Python wrapper (read-only, /root/reference) over a
`_C` in this GPU-less container and (b) to exercise dmesh_renderer_amd.sharding with the gloo backend.
The product never imports this module.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch as th

from oracle import oracle as O

_states = {}
_ids = itertools.count(1)
calls = []  # (name, [arg descriptors]) log for the drop-in tests


def _desc(a):
    if isinstance(a, th.Tensor):
        return ("tensor", str(a.dtype), tuple(a.shape), tuple(a.stride()), a.is_contiguous())
    return (type(a).__name__, a)


def _handle(st) -> th.Tensor:
    i = next(_ids)
    _states[i] = st
    return th.tensor([i], dtype=th.int64).view(th.uint8).clone()


def _state(buf: th.Tensor):
    return _states[int(buf.contiguous().view(th.int64)[0])]


def _np(t):
    return t.detach().contiguous().cpu().numpy()


def _scene(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, H, W, tets=None, ft=None, tf=None, seed=0, rows=(0, 0)):
    return O.Scene(_np(bg), _np(verts), _np(faces), _np(vc), _np(fo), _np(mv), _np(proj), _np(imv), _np(iproj),
                   _np(vd), _np(fi), H, W, tets=None if tets is None else _np(tets),
                   face_tets=None if ft is None else _np(ft), tet_faces=None if tf is None else _np(tf),
                   ray_random_seed=seed, rows=rows)


def render_tris(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, H, W, rows=(0, 0)):
    calls.append(("render_tris", [_desc(a) for a in (bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, H, W)]))
    sc = _scene(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, H, W, rows=rows)
    color, depth, st = O.tri_forward(sc)
    h = _handle((sc, st))
    e = th.empty(0, dtype=th.uint8)
    return st.num_rendered, th.from_numpy(color), th.from_numpy(depth), h, e.clone(), e.clone(), e.clone()


def render_tris_backward(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, gc, gd, R, pb, fb, bb, ib, rows=(0, 0)):
    calls.append(("render_tris_backward", [_desc(a) for a in (bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi,
                                                               gc, gd, R, pb, fb, bb, ib)]))
    sc, st = _state(pb)
    assert R == st.num_rendered
    g = O.tri_backward(sc, st, _np(gc), _np(gd))
    return tuple(th.from_numpy(g[k]) for k in ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense"))


def render_tets(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, tets, ft, tf, H, W, seed, rows=(0, 0)):
    calls.append(("render_tets", [_desc(a) for a in (bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, tets, ft, tf,
                                                      H, W, seed)]))
    sc = _scene(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, H, W, tets, ft, tf, seed, rows)
    color, depth, active, st = O.tet_forward(sc)
    h = _handle((sc, st))
    e = th.empty(0, dtype=th.uint8)
    return th.from_numpy(color), th.from_numpy(depth), th.from_numpy(active), h, e.clone(), e.clone(), e.clone()


def render_tets_backward(bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi, tets, ft, tf, gc, gd, pb, fb, bb, ib,
                         rows=(0, 0)):
    calls.append(("render_tets_backward", [_desc(a) for a in (bg, verts, faces, vc, fo, mv, proj, imv, iproj, vd, fi,
                                                               tets, ft, tf, gc, gd, pb, fb, bb, ib)]))
    sc, st = _state(pb)
    g = O.tet_backward(sc, st, _np(gc), _np(gd))
    return th.from_numpy(g["verts_color"]), th.from_numpy(g["faces_opacity"])
