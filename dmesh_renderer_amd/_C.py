"""`_C`: the four-function binding surface of the reference extension (ext.cpp:6-11),
implemented as thin PyTorch-ROCm glue over the C ABI of libdmesh_renderer_hip.so.

    render_tris            <- RasterizeTrianglesCUDA          (render.cu:29-132)
    render_tris_backward   <- RasterizeTrianglesBackwardCUDA  (render.cu:134-208)
    render_tets            <- RenderFTetsCUDA                 (render.cu:213-336)
    render_tets_backward   <- RenderFTetsBackwardCUDA         (render.cu:338-412)

Same positional arguments, same tuple arity, same dtypes/shapes, same error behaviour
(RuntimeError with the reference's messages).  PyTorch is used for device memory and the
current stream only; all compute is in the HIP library.  Extension over the reference: an
optional keyword `rows=(begin, end)` restricts a call to a band of tile rows (multi-GPU
tile-row sharding); the default renders everything.

There is no CPU path: tensors must live on a HIP device (`torch.device('cuda')`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch as th

from . import _lib

NUM_CHANNELS = 3  # cuda_*/config.h:4
SUPPORTS_FLAT_OUT = True  # render_tris_backward(flat_out=...), used by sharding.py


def _err(msg: str):
    raise RuntimeError(msg)


def _f32(t: th.Tensor, name: str) -> th.Tensor:
    # render.cu:113-129: `.contiguous().data<float>()` throws for any other dtype
    if t.dtype != th.float32:
        _err(f"expected scalar type Float but found {t.dtype} ({name})")
    return t.contiguous()


def _mat(t: th.Tensor, name: str):
    """[B,4,4] matrix -> (tensor to keep alive, transposed-storage flag).  render.cu:117-120 makes the
    `.transpose(1, 2)` views of the wrapper contiguous with four copy kernels per call; the library reads
    that storage in place instead (dmr_scene.mats_transposed)."""
    if t.dtype != th.float32:
        _err(f"expected scalar type Float but found {t.dtype} ({name})")
    if t.is_contiguous():
        return t, 0
    if t.dim() == 3 and t.transpose(1, 2).is_contiguous():
        return t, 1
    return t.contiguous(), 0


def _i32(t: th.Tensor, name: str) -> th.Tensor:
    if t.dtype != th.int32:
        _err(f"expected scalar type Int but found {t.dtype} ({name})")
    return t.contiguous()


def _check_common(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats,
                  verts_depth, faces_intense, tet: bool):
    # messages: render.cu:49-79 (tri) and :237-267 (tet)
    if verts.dim() != 2 or verts.size(1) != 3:
        _err("verts must have dimensions (num_points, 3)")
    if faces.dim() != 2 or faces.size(1) != 3:
        _err("faces must have dimensions (num_faces, 3)")
    if tet:
        if verts_color.dim() != 2 or verts_color.size(0) != verts.size(0) or verts_color.size(1) != 3:
            _err("vert_color must have dimensions (num_verts, 3)")
        if faces_opacity.dim() != 1 or faces_opacity.size(0) != faces.size(0):
            _err("face_opacity must have dimensions (num_faces)")
    else:
        if verts_color.dim() != 2 or verts_color.size(0) != verts.size(0):
            _err("vert color must have dimensions (num_points, N)")
        if verts_color.size(1) != NUM_CHANNELS:  # Q15: the kernels assume 3 channels
            _err("vert color must have dimensions (num_points, 3)")
        if faces_opacity.dim() != 1 or faces_opacity.size(0) != faces.size(0):
            _err("face opacity must have dimensions (num_faces,)")
    bdim = "batch_size" if tet else "B"
    for m, n in ((mv_mats, "mv_mats"), (proj_mats, "proj_mats"), (inv_mv_mats, "inv_mv_mats"),
                 (inv_proj_mats, "inv_proj_mats")):
        if m.dim() != 3 or m.size(1) != 4 or m.size(2) != 4:
            _err(f"{n} must have dimensions ({bdim}, 4, 4)")
    if verts_depth.dim() != 2 or verts_depth.size(1) != verts.size(0):
        _err("verts_depth must have dimensions (batch_size, num_verts)" if tet
             else "verts_depth must have dimensions (B, num_points,)")
    if faces_intense.dim() != 2 or faces_intense.size(1) != faces.size(0):
        _err("faces_intense must have dimensions (batch_size, num_faces)" if tet
             else "faces_intense must have dimensions (B, num_faces,)")
    B = mv_mats.size(0)
    for m, n in ((proj_mats, "proj_mats"), (inv_mv_mats, "inv_mv_mats"), (inv_proj_mats, "inv_proj_mats"),
                 (verts_depth, "verts_depth"), (faces_intense, "faces_intense")):
        if m.size(0) != B:  # the reference would read out of bounds here
            _err(f"{n} must have the batch size of mv_mats ({B})")


def _device_of(verts: th.Tensor) -> th.device:
    if not verts.is_cuda:
        _err("dmesh_renderer_amd has no CPU path: tensors must be on a HIP device "
             "(the reference allocates on torch::kCUDA unconditionally, render.cu:91-96)")
    return verts.device


class _Call:
    """Owns the contiguous input tensors, the dmr_scene struct and the scratch allocator of one call."""

    def __init__(self, dev, bg, verts, faces, verts_color, faces_opacity, mv, proj, inv_mv, inv_proj, verts_depth,
                 faces_intense, H, W, tets=None, face_tets=None, tet_faces=None, seed=0, rows=(0, 0)):
        self.dev = dev
        (mv, f0), (proj, f1), (inv_mv, f2), (inv_proj, f3) = (_mat(mv, "mv_mats"), _mat(proj, "proj_mats"),
                                                             _mat(inv_mv, "inv_mv_mats"), _mat(inv_proj, "inv_proj_mats"))
        k = self.keep = {
            "bg": _f32(bg, "background"), "verts": _f32(verts, "verts"), "faces": _i32(faces, "faces"),
            "vc": _f32(verts_color, "verts_color"), "fo": _f32(faces_opacity, "faces_opacity"),
            "mv": mv, "proj": proj, "imv": inv_mv, "iproj": inv_proj,
            "vd": _f32(verts_depth, "verts_depth"), "fi": _f32(faces_intense, "faces_intense"),
        }
        if bg.numel() < NUM_CHANNELS:
            _err("background must have 3 channels")
        if tets is not None:
            k["tets"], k["ft"], k["tf"] = _i32(tets, "tets"), _i32(face_tets, "face_tets"), _i32(tet_faces, "tet_faces")
        for n, t in k.items():
            if t.device != dev:
                _err(f"all tensors must be on {dev} ({n} is on {t.device})")
        p = lambda n: k[n].data_ptr() if n in k and k[n].numel() else None
        self.B, self.P, self.F = mv.size(0), verts.size(0), faces.size(0)
        self.T = 0 if tets is None else tets.size(0)
        self.H, self.W = int(H), int(W)
        self.scene = _lib.Scene(self.B, self.P, self.F, self.T, self.W, self.H,
                                p("bg"), p("verts"), p("faces"), p("vc"), p("fo"),
                                p("mv"), p("proj"), p("imv"), p("iproj"), p("vd"), p("fi"),
                                p("tets"), p("ft"), p("tf"), int(seed), int(rows[0]), int(rows[1]),
                                f0 | (f1 << 1) | (f2 << 2) | (f3 << 3))
        self.buffers: Dict[int, th.Tensor] = {}

        def alloc(_ctx, which, nbytes):
            try:
                t = th.empty(max(int(nbytes), 1), dtype=th.uint8, device=dev)
                self.buffers[which] = t
                return t.data_ptr()
            except Exception:  # reported by the library as an allocation failure
                return None

        self.alloc = _lib.ALLOC_FN(alloc)

    def buf(self, which: int) -> th.Tensor:
        t = self.buffers.get(which)
        return t if t is not None else th.empty(0, dtype=th.uint8, device=self.dev)

    def stream(self):
        return C.c_void_p(th.cuda.current_stream(self.dev).cuda_stream)


class _on_device:
    """`with th.cuda.device(dev)` when dev is not already the current device (the context manager costs ~10 us of a
    small step; the common case is one device per process)."""

    def __init__(self, dev):
        self.ctx = None if th.cuda.current_device() == dev.index else th.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _ptr(t: th.Tensor):
    return t.data_ptr() if t.numel() else None


def _raise_lib():
    _err(_lib.last_error())


def render_tris(background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                inv_proj_mats, verts_depth, faces_intense, image_height, image_width, rows=(0, 0), fill_outside=True):
    """-> (num_rendered:int, color [B,3,H,W], depth [B,1,H,W], pointBuffer, faceBuffer, binningBuffer, imgBuffer)

    Extensions for the sharded path: `rows=(begin, end)` renders a band of 16-pixel tile rows; pixels outside it are
    zero, or left uninitialised with `fill_outside=False` (a rank that only ever reads its own rows saves the fill)."""
    lib = _lib.load()
    _check_common(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats,
                  verts_depth, faces_intense, tet=False)
    dev = _device_of(verts)
    with _on_device(dev):
        call = _Call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                     inv_proj_mats, verts_depth, faces_intense, image_height, image_width, rows=rows)
        # the kernels write every pixel of the rendered rows; zero-fill (render.cu:88-89) is only needed when
        # nothing is launched (P == 0 / F == 0, render.cu:105) or when a band leaves rows untouched
        full = (tuple(rows) == (0, 0) or not fill_outside) and call.P > 0 and call.F > 0
        alloc_img = th.empty if full else th.zeros
        color = alloc_img((call.B, NUM_CHANNELS, call.H, call.W), dtype=th.float32, device=dev)
        depth = alloc_img((call.B, 1, call.H, call.W), dtype=th.float32, device=dev)
        rendered = C.c_int(0)
        rc = lib.dmr_tri_forward(C.byref(call.scene), color.data_ptr(), depth.data_ptr(), call.alloc, None,
                                 call.stream(), C.byref(rendered))
        if rc:
            _raise_lib()
    return (rendered.value, color, depth, call.buf(_lib.BUF_POINT), call.buf(_lib.BUF_FACE),
            call.buf(_lib.BUF_BINNING), call.buf(_lib.BUF_IMAGE))


def render_tris_backward(background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                         inv_proj_mats, verts_depth, faces_intense, dL_dout_color, dL_dout_depth, R,
                         pointBuffer, faceBuffer, binningBuffer, imageBuffer, rows=(0, 0), flat_out=None):
    """-> (dL_dverts [P,3], dL_dvcolor [P,3], dL_dfopacity [F], dL_dvdepth [B,P], dL_dfintense [B,F])

    Extension for the sharded path: `flat_out`, a float32 HIP tensor of 6P + F + B(P + F) elements, receives the five
    gradients back to back ([3P | 3P | F | BP | BF], the layout of the one all-reduce in sharding.py) and the
    returned tensors are views into it -- no concatenation kernel before the collective."""
    lib = _lib.load()
    dev = _device_of(verts)
    H, W = dL_dout_color.size(2), dL_dout_color.size(3)  # render.cu:163-164
    with _on_device(dev):
        call = _Call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                     inv_proj_mats, verts_depth, faces_intense, H, W, rows=rows)
        gc = _f32(dL_dout_color, "dL_dout_color")  # may arrive non-contiguous / expanded (render.cu:197-198)
        gd = _f32(dL_dout_depth, "dL_dout_depth")
        B, P, F = call.B, call.P, call.F
        if flat_out is None:
            g_verts = th.empty((P, 3), dtype=th.float32, device=dev)
            g_vcolor = th.empty((P, NUM_CHANNELS), dtype=th.float32, device=dev)
            g_fop = th.empty((F,), dtype=th.float32, device=dev)
            g_vdepth = th.empty((B, P), dtype=th.float32, device=dev)
            g_fint = th.empty((B, F), dtype=th.float32, device=dev)
        else:
            sizes = (3 * P, 3 * P, F, B * P, B * F)
            if flat_out.dtype != th.float32 or flat_out.device != dev or not flat_out.is_contiguous() \
                    or flat_out.numel() != sum(sizes):
                _err(f"flat_out must be a contiguous float32 tensor of {sum(sizes)} elements on {dev}")
            parts = th.split(flat_out.view(-1), sizes)
            g_verts, g_vcolor, g_fop = parts[0].view(P, 3), parts[1].view(P, NUM_CHANNELS), parts[2]
            g_vdepth, g_fint = parts[3].view(B, P), parts[4].view(B, F)
        bufs = [b.contiguous() for b in (pointBuffer, faceBuffer, binningBuffer, imageBuffer)]
        rc = lib.dmr_tri_backward(C.byref(call.scene), _ptr(gc), _ptr(gd), int(R), *[_ptr(b) for b in bufs],
                                  _ptr(g_verts), _ptr(g_vcolor), _ptr(g_fop), _ptr(g_vdepth), _ptr(g_fint),
                                  call.alloc, None, call.stream())
        if rc:
            _raise_lib()
    return g_verts, g_vcolor, g_fop, g_vdepth, g_fint


def _check_tets(faces, tets, face_tets, tet_faces):
    # render.cu:269-277
    if tets.dim() != 2 or tets.size(1) != 4:
        _err("tets must have dimensions (num_tets, 4)")
    if face_tets.dim() != 2 or face_tets.size(0) != faces.size(0) or face_tets.size(1) != 2:
        _err("face_tets must have dimensions (num_faces, 2)")
    if tet_faces.dim() != 2 or tet_faces.size(0) != tets.size(0) or tet_faces.size(1) != 4:
        _err("tet_faces must have dimensions (num_tets, 4)")


def render_tets(background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                inv_proj_mats, verts_depth, faces_intense, tets, face_tets, tet_faces, image_height, image_width,
                ray_random_seed, rows=(0, 0)):
    """-> (color [B,3,H,W], depth [B,1,H,W], active f32 [B,H,W], pointBuffer, faceBuffer, binningBuffer, imgBuffer)"""
    lib = _lib.load()
    _check_common(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats,
                  verts_depth, faces_intense, tet=True)
    _check_tets(faces, tets, face_tets, tet_faces)
    dev = _device_of(verts)
    with _on_device(dev):
        call = _Call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                     inv_proj_mats, verts_depth, faces_intense, image_height, image_width,
                     tets=tets, face_tets=face_tets, tet_faces=tet_faces, seed=ray_random_seed, rows=rows)
        # k_tet_forward writes every pixel of the rendered rows (background where the march fails); zero-fill
        # (render.cu:287-290) is only needed when a band leaves rows untouched or nothing is launched
        full = tuple(rows) == (0, 0) and call.P > 0 and call.F > 0
        alloc_img = th.empty if full else th.zeros
        color = alloc_img((call.B, NUM_CHANNELS, call.H, call.W), dtype=th.float32, device=dev)
        depth = alloc_img((call.B, 1, call.H, call.W), dtype=th.float32, device=dev)
        active = alloc_img((call.B, call.H, call.W), dtype=th.float32, device=dev)
        rendered = C.c_int(0)
        rc = lib.dmr_tet_forward(C.byref(call.scene), color.data_ptr(), depth.data_ptr(), active.data_ptr(),
                                 call.alloc, None, call.stream(), C.byref(rendered))
        if rc:
            _raise_lib()
    return (color, depth, active, call.buf(_lib.BUF_POINT), call.buf(_lib.BUF_FACE),
            call.buf(_lib.BUF_BINNING), call.buf(_lib.BUF_IMAGE))


def render_tets_backward(background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                         inv_proj_mats, verts_depth, faces_intense, tets, face_tets, tet_faces, grad_color,
                         grad_depth, pointBuffer, faceBuffer, binningBuffer, imageBuffer, rows=(0, 0), flat_out=None):
    """-> (dL_dverts_color [P,3], dL_dfaces_opacity [F])

    Extension for the sharded path: `flat_out`, a float32 HIP tensor of 3P + F elements, receives both gradients back
    to back (the layout of the one all-reduce in sharding.py); the returned tensors are views into it."""
    lib = _lib.load()
    dev = _device_of(verts)
    H, W = grad_color.size(2), grad_color.size(3)  # render.cu:371-372
    with _on_device(dev):
        call = _Call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats,
                     inv_proj_mats, verts_depth, faces_intense, H, W,
                     tets=tets, face_tets=face_tets, tet_faces=tet_faces, seed=0, rows=rows)
        gc, gd = _f32(grad_color, "grad_color"), _f32(grad_depth, "grad_depth")
        if flat_out is None:
            g_vcolor = th.empty((call.P, 3), dtype=th.float32, device=dev)
            g_fop = th.empty((call.F,), dtype=th.float32, device=dev)
        else:
            n = 3 * call.P + call.F
            if flat_out.dtype != th.float32 or flat_out.device != dev or not flat_out.is_contiguous() or flat_out.numel() != n:
                _err(f"flat_out must be a contiguous float32 tensor of {n} elements on {dev}")
            g_vcolor, g_fop = flat_out.view(-1)[:3 * call.P].view(call.P, 3), flat_out.view(-1)[3 * call.P:]
        bufs = [b.contiguous() for b in (pointBuffer, faceBuffer, binningBuffer, imageBuffer)]
        rc = lib.dmr_tet_backward(C.byref(call.scene), _ptr(gc), _ptr(gd), *[_ptr(b) for b in bufs],
                                  _ptr(g_vcolor), _ptr(g_fop), call.alloc, None, call.stream())
        if rc:
            _raise_lib()
    return g_vcolor, g_fop


def invert_mats(*mats: th.Tensor):
    """th.inverse of [B,4,4] float32 HIP tensors with one small library kernel each (dmr_invert_mats: adjugate in
    double precision).  Extension over the reference's `_C` (its wrapper calls th.inverse twice per forward,
    dmesh_renderer/__init__.py:62-63: two batched LU factorisations, ~0.12 ms of small kernels on the GPU).
    Returns contiguous tensors."""
    lib = _lib.load()
    dev = _device_of(mats[0])
    res = []
    with _on_device(dev):
        stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)
        for m in mats:
            if m.dim() != 3 or m.size(1) != 4 or m.size(2) != 4:
                _err("matrices must have dimensions (B, 4, 4)")
            t, flag = _mat(m, "matrix")
            out = th.empty((t.size(0), 4, 4), dtype=th.float32, device=dev)
            if t.size(0) and lib.dmr_invert_mats(t.data_ptr(), t.size(0), flag, out.data_ptr(), stream):
                _raise_lib()
            res.append(out)
    return tuple(res)


def export(name: str, call_args: Tuple, is_tet: bool, num_rendered: int, buffers, H: int, W: int,
           dtype=th.float32) -> th.Tensor:
    """Parity/debug helper: copy one forward intermediate out of the scratch buffers
    (dmr_export).  `call_args` are the 11 (tri) / 14 (tet) leading tensors of render_*."""
    lib = _lib.load()
    dev = call_args[1].device
    with _on_device(dev):
        tet_kw = dict(tets=call_args[11], face_tets=call_args[12], tet_faces=call_args[13]) if is_tet else {}
        call = _Call(dev, *call_args[:11], H, W, **tet_kw)
        bufs = [_ptr(b) for b in buffers]
        n = lib.dmr_export(C.byref(call.scene), int(is_tet), int(num_rendered), name.encode(), *bufs, None, 0,
                           call.stream())
        if n < 0:
            _raise_lib()
        out = th.empty(max(n, 1), dtype=th.uint8, device=dev)
        lib.dmr_export(C.byref(call.scene), int(is_tet), int(num_rendered), name.encode(), *bufs,
                       out.data_ptr(), n, call.stream())
        return out[:n].view(dtype)
