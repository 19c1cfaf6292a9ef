"""ctypes view of the C ABI in include/dmesh_renderer_amd.h -- TEST INFRASTRUCTURE (tests/test_capi_cpu.py): the
product's binding is the compiled module dmesh_renderer_amd._C (csrc/dmr_torch.cpp); this table restates the header
in a second, independent form so that the library's exports, the struct layout and the version can be checked without
a GPU, and shows what a non-PyTorch host would bind.  torch is imported first so that the library
binds to the HIP runtime already loaded by PyTorch-ROCm (one runtime per process).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmesh_renderer_amd")
LIB_PATH = os.environ.get("DMR_LIBRARY") or os.path.join(HERE, "libdmesh_renderer_hip.so")
ABI_VERSION = 4

BUF_POINT, BUF_FACE, BUF_BINNING, BUF_IMAGE, BUF_WORK = range(5)
NUM_STAGES = 12

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)


class Scene(C.Structure):
    """struct dmr_scene"""
    _fields_ = [
        ("B", C.c_int32), ("P", C.c_int32), ("F", C.c_int32), ("T", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("background", C.c_void_p), ("verts", C.c_void_p), ("faces", C.c_void_p),
        ("verts_color", C.c_void_p), ("faces_opacity", C.c_void_p),
        ("mv_mats", C.c_void_p), ("proj_mats", C.c_void_p), ("inv_mv_mats", C.c_void_p), ("inv_proj_mats", C.c_void_p),
        ("verts_depth", C.c_void_p), ("faces_intense", C.c_void_p),
        ("tets", C.c_void_p), ("face_tets", C.c_void_p), ("tet_faces", C.c_void_p),
        ("ray_random_seed", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("mats_transposed", C.c_int32), ("flags", C.c_int32),
    ]


EXPORTS = {
    "dmr_tri_forward": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p, ALLOC_FN, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_int)]),
    "dmr_tri_backward": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
                         + [C.c_void_p] * 5 + [ALLOC_FN, C.c_void_p, C.c_void_p]),
    "dmr_tet_forward": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p, ALLOC_FN, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int)]),
    "dmr_tet_backward": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p] + [C.c_void_p] * 4
                         + [C.c_void_p] * 2 + [ALLOC_FN, C.c_void_p, C.c_void_p]),
    "dmr_export": (C.c_int64, [C.POINTER(Scene), C.c_int, C.c_int, C.c_char_p] + [C.c_void_p] * 4
                   + [C.c_void_p, C.c_int64, C.c_void_p]),
    "dmr_invert_mats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dmr_redo_count": (C.c_uint64, []),
    "dmr_profile_enable": (None, [C.c_uint32]),
    "dmr_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dmr_stage_name": (C.c_char_p, [C.c_int]),
    "dmr_last_error": (C.c_char_p, []),
    "dmr_overflowed": (C.c_int, [C.c_int, C.c_int]),
    "dmr_abi_version": (C.c_int, []),
    "dmr_build_arch": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the library and bind every symbol of the ABI; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python -m dmesh_renderer_amd.build`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dmr_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.dmr_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().dmr_last_error().decode(errors="replace")
