"""ctypes front-end of the CPU oracle (oracle/libdmr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (dmesh_renderer_amd) must never import this.
PARITY UNPINNED at kernel level -- see oracle/dmr_oracle.h.

Inputs follow the `_C` convention of the reference (render.cu:29-132): matrices are the
already transposed, column-major [B,4,4] tensors (m[4*col+row] once made contiguous).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdmr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dmr_oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "dmr_oracle.h")))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdmr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Scene(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("P", C.c_int), ("F", C.c_int), ("T", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("background", C.c_void_p), ("verts", C.c_void_p), ("faces", C.c_void_p),
        ("verts_color", C.c_void_p), ("faces_opacity", C.c_void_p),
        ("mv_mats", C.c_void_p), ("proj_mats", C.c_void_p), ("inv_mv_mats", C.c_void_p), ("inv_proj_mats", C.c_void_p),
        ("verts_depth", C.c_void_p), ("faces_intense", C.c_void_p),
        ("tets", C.c_void_p), ("face_tets", C.c_void_p), ("tet_faces", C.c_void_p),
        ("ray_random_seed", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.dmro_tri_forward.restype = C.c_void_p
        L.dmro_tri_forward.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p]
        L.dmro_tri_backward.restype = C.c_int
        L.dmro_tri_backward.argtypes = [C.POINTER(_Scene), C.c_void_p] + [C.c_void_p] * 7
        L.dmro_tet_forward.restype = C.c_void_p
        L.dmro_tet_forward.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.dmro_tet_backward.restype = C.c_int
        L.dmro_tet_backward.argtypes = [C.POINTER(_Scene), C.c_void_p] + [C.c_void_p] * 4
        L.dmro_num_rendered.restype = C.c_int64
        L.dmro_num_rendered.argtypes = [C.c_void_p]
        L.dmro_get.restype = C.c_int64
        L.dmro_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.dmro_free.argtypes = [C.c_void_p]
        L.dmro_last_error.restype = C.c_char_p
        L.dmro_set_tri_grad_f64.restype = None
        L.dmro_set_tri_grad_f64.argtypes = [C.c_int]
        L.dmro_num_threads.restype = C.c_int
        L.dmro_in_tri.restype = C.c_int
        L.dmro_in_tri.argtypes = [C.c_float] * 8
        L.dmro_clamp_bary_uv.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.dmro_ray_tri.restype = C.c_int
        L.dmro_ray_tri.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.dmro_ndc2pix.restype = C.c_float
        L.dmro_ndc2pix.argtypes = [C.c_float, C.c_int]
        L.dmro_pix2ndc.restype = C.c_float
        L.dmro_pix2ndc.argtypes = [C.c_float, C.c_int]
        L.dmro_rect_from_tri.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p]
        L.dmro_higher_msb.restype = C.c_uint32
        L.dmro_higher_msb.argtypes = [C.c_uint32]
        _lib = L
    return _lib


_DTYPES = {
    "ndc": np.float32, "image": np.float32, "depths": np.float32, "min_depths": np.float32,
    "max_depths": np.float32, "tiles_touched": np.uint32, "face_offsets": np.uint32,
    "keys": np.uint64, "values": np.uint32, "ranges": np.uint32, "ray_o": np.float32,
    "ray_d": np.float32, "final_T": np.float32, "final_prev_T": np.float32, "n_contrib": np.uint32,
    "first_face": np.int32, "first_tet": np.int32, "last_face": np.int32, "last_tet": np.int32,
    "is_active": np.uint8,
}


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class Scene:
    """Holds contiguous numpy copies of the `_C`-convention inputs and the C struct."""

    def __init__(self, bg, verts, faces, verts_color, faces_opacity, mv, proj, inv_mv, inv_proj,
                 verts_depth, faces_intense, H, W, tets=None, face_tets=None, tet_faces=None,
                 ray_random_seed=0, rows=(0, 0)):
        self.bg = _f32(bg)
        self.verts = _f32(verts).reshape(-1, 3)
        self.faces = _i32(faces).reshape(-1, 3)
        self.verts_color = _f32(verts_color).reshape(-1, 3)
        self.faces_opacity = _f32(faces_opacity).reshape(-1)
        self.mv, self.proj = _f32(mv).reshape(-1, 16), _f32(proj).reshape(-1, 16)
        self.inv_mv, self.inv_proj = _f32(inv_mv).reshape(-1, 16), _f32(inv_proj).reshape(-1, 16)
        self.B, self.P, self.F = self.mv.shape[0], self.verts.shape[0], self.faces.shape[0]
        self.verts_depth = _f32(verts_depth).reshape(self.B, self.P)
        self.faces_intense = _f32(faces_intense).reshape(self.B, self.F)
        self.H, self.W = int(H), int(W)
        self.tets = _i32(tets).reshape(-1, 4) if tets is not None else None
        self.face_tets = _i32(face_tets).reshape(-1, 2) if face_tets is not None else None
        self.tet_faces = _i32(tet_faces).reshape(-1, 4) if tet_faces is not None else None
        self.T = 0 if self.tets is None else self.tets.shape[0]
        p = lambda a: None if a is None else a.ctypes.data
        self.c = _Scene(self.B, self.P, self.F, self.T, self.W, self.H,
                        p(self.bg), p(self.verts), p(self.faces), p(self.verts_color), p(self.faces_opacity),
                        p(self.mv), p(self.proj), p(self.inv_mv), p(self.inv_proj),
                        p(self.verts_depth), p(self.faces_intense),
                        p(self.tets), p(self.face_tets), p(self.tet_faces),
                        int(ray_random_seed), int(rows[0]), int(rows[1]))


class State:
    def __init__(self, handle, scene: Scene):
        self.h = handle
        self.scene = scene

    @property
    def num_rendered(self) -> int:
        return int(lib().dmro_num_rendered(self.h))

    def get(self, name: str) -> np.ndarray:
        n = lib().dmro_get(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n // np.dtype(_DTYPES[name]).itemsize, dtype=_DTYPES[name])
        if n:
            lib().dmro_get(self.h, name.encode(), out.ctypes.data, n)
        return out

    def close(self):
        if self.h:
            lib().dmro_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _err():
    return RuntimeError(lib().dmro_last_error().decode())


def tri_forward(sc: Scene):
    color = np.empty((sc.B, 3, sc.H, sc.W), dtype=np.float32)
    depth = np.empty((sc.B, 1, sc.H, sc.W), dtype=np.float32)
    h = lib().dmro_tri_forward(C.byref(sc.c), color.ctypes.data, depth.ctypes.data)
    if not h:
        raise _err()
    return color, depth, State(h, sc)


def tri_backward(sc: Scene, st: State, dL_dcolor, dL_ddepth, verts_grad_f64: bool = False) -> Dict[str, np.ndarray]:
    """verts_grad_f64 (noise measurement only): the per-pair vertex-position gradient is evaluated in double."""
    gc, gd = _f32(dL_dcolor), _f32(dL_ddepth)
    lib().dmro_set_tri_grad_f64(1 if verts_grad_f64 else 0)
    out = {
        "verts": np.empty((sc.P, 3), np.float32), "verts_color": np.empty((sc.P, 3), np.float32),
        "faces_opacity": np.empty((sc.F,), np.float32), "verts_depth": np.empty((sc.B, sc.P), np.float32),
        "faces_intense": np.empty((sc.B, sc.F), np.float32),
    }
    rc = lib().dmro_tri_backward(C.byref(sc.c), st.h, gc.ctypes.data, gd.ctypes.data,
                                 *[out[k].ctypes.data for k in ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")])
    lib().dmro_set_tri_grad_f64(0)
    if rc:
        raise _err()
    return out


def tet_forward(sc: Scene):
    color = np.empty((sc.B, 3, sc.H, sc.W), dtype=np.float32)
    depth = np.empty((sc.B, 1, sc.H, sc.W), dtype=np.float32)
    active = np.empty((sc.B, sc.H, sc.W), dtype=np.float32)
    h = lib().dmro_tet_forward(C.byref(sc.c), color.ctypes.data, depth.ctypes.data, active.ctypes.data)
    if not h:
        raise _err()
    return color, depth, active, State(h, sc)


def tet_backward(sc: Scene, st: State, dL_dcolor, dL_ddepth) -> Dict[str, np.ndarray]:
    gc, gd = _f32(dL_dcolor), _f32(dL_ddepth)
    out = {"verts_color": np.empty((sc.P, 3), np.float32), "faces_opacity": np.empty((sc.F,), np.float32)}
    rc = lib().dmro_tet_backward(C.byref(sc.c), st.h, gc.ctypes.data, gd.ctypes.data,
                                 out["verts_color"].ctypes.data, out["faces_opacity"].ctypes.data)
    if rc:
        raise _err()
    return out


def scene_from_module_inputs(d: dict, H: int, W: int, rows=(0, 0), seed: int = 0) -> Scene:
    """Build a `_C`-convention Scene from row-major module inputs (dmesh_renderer_amd.scenes):
    transposes mv/proj and inverts them, as reference __init__.py:62-63,219-220 does."""
    import torch as th
    mv_t = d["mv_mats"].transpose(1, 2)
    proj_t = d["proj_mats"].transpose(1, 2)
    inv_mv = th.inverse(mv_t)
    inv_proj = th.inverse(proj_t)
    c = lambda t: t.contiguous().numpy()
    return Scene(c(d["bg"]), c(d["verts"]), c(d["faces"]), c(d["verts_color"]), c(d["faces_opacity"]),
                 c(mv_t), c(proj_t), c(inv_mv), c(inv_proj), c(d["verts_depth"]), c(d["faces_intense"]),
                 H, W, tets=c(d["tets"]) if "tets" in d else None,
                 face_tets=c(d["face_tets"]) if "face_tets" in d else None,
                 tet_faces=c(d["tet_faces"]) if "tet_faces" in d else None,
                 ray_random_seed=seed, rows=rows)
