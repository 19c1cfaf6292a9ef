"""dmesh_renderer_amd -- MI355X-native drop-in for the hot path of SonSang/dmesh_renderer.

Public surface = the reference package's (dmesh_renderer/__init__.py):
    TriRenderSettings, render_tri, TriRenderer      (:13-16, :18-43, :172-225)
    TetRenderSettings, render_tet, TetRenderer      (:237-241, :243-275, :426-488)
with identical argument order, dtypes, shapes, outputs and gradient routing.  The compute
lives in libdmesh_renderer_hip.so (hand-written gfx950 kernels) behind `_C`.

Conventions kept from the reference: the Modules receive ROW-major [B,4,4] matrices and
pass transposed views down (:219-220, :476-477); the autograd Functions invert those
(:62-63, :298-299); the tri Module only casts `faces`, the tet Module casts everything (Q23).
"""
from __future__ import annotations

from typing import NamedTuple, Tuple

import torch as th

class _NotBuilt:
    """Stands in for `_C` while the native pieces are missing, so that `dmesh_renderer_amd.build` itself stays
    importable; any use of the renderer raises -- there is no CPU or Python fallback."""

    def __init__(self, cause):
        self._cause = cause

    def __getattr__(self, name):
        raise ImportError(f"dmesh_renderer_amd._C cannot be imported ({self._cause}). Build the native pieces first: "
                          "`python -m dmesh_renderer_amd.build` (hipcc + g++, in-tree); there is no CPU or Python fallback.")


try:
    from . import _C  # the compiled binding (csrc/dmr_torch.cpp) over libdmesh_renderer_hip.so
except ImportError as _e:  # not built yet, or its HIP library is missing / of another ABI version
    _C = _NotBuilt(_e)

__all__ = ["TriRenderSettings", "render_tri", "TriRenderer", "TetRenderSettings", "render_tet", "TetRenderer"]


class TriRenderSettings(NamedTuple):
    image_height: int
    image_width: int
    bg: th.Tensor


class TetRenderSettings(NamedTuple):
    image_height: int
    image_width: int
    bg: th.Tensor
    ray_random_seed: int


def _with_inverses(mv_mats: th.Tensor, proj_mats: th.Tensor) -> Tuple[th.Tensor, ...]:
    """(mv, proj, mv^-1, proj^-1), reference :62-63.  On a HIP device the two inverses come from one library kernel
    (_C.invert_mats) instead of two th.inverse calls; anything else (CPU tensors in the wrapper tests, other
    dtypes) takes th.inverse like the reference."""
    if mv_mats.is_cuda and proj_mats.is_cuda and mv_mats.dtype == th.float32 and proj_mats.dtype == th.float32 \
            and mv_mats.dim() == 3 and proj_mats.dim() == 3:
        return (mv_mats, proj_mats) + _C.invert_mats(mv_mats, proj_mats)
    return mv_mats, proj_mats, th.inverse(mv_mats), th.inverse(proj_mats)


class _TriFn(th.autograd.Function):
    """Inputs: verts, faces, verts_color, faces_opacity, mv^T, proj^T, verts_depth, faces_intense,
    settings, rows.  Gradients flow to verts, verts_color, faces_opacity, verts_depth, faces_intense."""

    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                settings: TriRenderSettings, rows):
        cams = _with_inverses(mv_mats, proj_mats)
        geom = (verts, faces, verts_color, faces_opacity)
        try:
            out = _C.render_tris(settings.bg, *geom, *cams, verts_depth, faces_intense,
                                 settings.image_height, settings.image_width, rows=rows)
        except Exception as ex:
            print("\nAn error occured in forward.")
            print(ex)
            raise
        num_rendered, color, depth = out[0], out[1], out[2]
        ctx.settings, ctx.rows, ctx.num_rendered = settings, rows, num_rendered
        ctx.save_for_backward(*geom, *cams, verts_depth, faces_intense, *out[3:7])
        return color, depth

    @staticmethod
    def backward(ctx, grad_color, grad_depth):
        saved = ctx.saved_tensors
        inputs, scratch = saved[:10], saved[10:14]
        try:
            g = _C.render_tris_backward(ctx.settings.bg, *inputs, grad_color, grad_depth, ctx.num_rendered,
                                        *scratch, rows=ctx.rows)
        except Exception:
            print("\nAn error occured in backward.\n")
            raise
        g_verts, g_vcolor, g_fopacity, g_vdepth, g_fintense = g
        return g_verts, None, g_vcolor, g_fopacity, None, None, g_vdepth, g_fintense, None, None


class _TetFn(th.autograd.Function):
    """Gradients flow to verts_color and faces_opacity only (reference :407-422)."""

    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                tets, face_tets, tet_faces, settings: TetRenderSettings, rows):
        cams = _with_inverses(mv_mats, proj_mats)
        geom = (verts, faces, verts_color, faces_opacity)
        topo = (tets, face_tets, tet_faces)
        try:
            out = _C.render_tets(settings.bg, *geom, *cams, verts_depth, faces_intense, *topo,
                                 settings.image_height, settings.image_width, settings.ray_random_seed, rows=rows)
        except Exception:
            print("\nAn error occured in forward.")
            raise
        color, depth, active = out[0], out[1], out[2] > 0.5  # bool mask, reference :333
        ctx.settings, ctx.rows = settings, rows
        ctx.save_for_backward(*geom, *cams, verts_depth, faces_intense, *topo, *out[3:7])
        ctx.mark_non_differentiable(active)
        return color, depth, active

    @staticmethod
    def backward(ctx, grad_color, grad_depth, _grad_active):
        saved = ctx.saved_tensors
        inputs, scratch = saved[:13], saved[13:17]
        try:
            g_vcolor, g_fopacity = _C.render_tets_backward(ctx.settings.bg, *inputs, grad_color, grad_depth,
                                                           *scratch, rows=ctx.rows)
        except Exception:
            print("\nAn error occured in backward.\n")
            raise
        return (None, None, g_vcolor, g_fopacity) + (None,) * 9


def render_tri(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
               render_settings: TriRenderSettings, rows=(0, 0)):
    """Functional form (reference :18-43).  mv_mats / proj_mats are the TRANSPOSED matrices."""
    return _TriFn.apply(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth,
                        faces_intense, render_settings, tuple(rows))


def render_tet(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
               tets, face_tets, tet_faces, render_settings: TetRenderSettings, rows=(0, 0)):
    """Functional form (reference :243-275).  mv_mats / proj_mats are the TRANSPOSED matrices."""
    return _TetFn.apply(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth,
                        faces_intense, tets, face_tets, tet_faces, render_settings, tuple(rows))


class TriRenderer(th.nn.Module):
    """Renderer for (semi-transparent) triangles: depth-sorted front-to-back compositing of every
    triangle of a tile; no exact per-pixel depth test (reference README.md:3).

    forward(verts [P,3], faces [F,3], verts_color [P,3], faces_opacity [F],
            mv_mats [B,4,4], proj_mats [B,4,4], verts_depth [B,P], faces_intense [B,F])
        -> color [B,3,H,W], depth [B,1,H,W]
    """

    def __init__(self, render_settings: TriRenderSettings):
        super().__init__()
        self.render_settings = render_settings

    def forward(self, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense):
        return render_tri(verts, faces.to(dtype=th.int32), verts_color, faces_opacity,
                          mv_mats.transpose(1, 2), proj_mats.transpose(1, 2), verts_depth, faces_intense,
                          self.render_settings)


class TetRenderer(th.nn.Module):
    """Renderer for the faces of a compact set of tetrahedra: the ray is marched tet to tet, so
    faces are composited in exact depth order; gradients reach verts_color and faces_opacity only
    (reference README.md:4).

    forward(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
            tets [T,4], face_tets [F,2] (-1 = none), tet_faces [T,4])
        -> color [B,3,H,W], depth [B,1,H,W], active bool [B,H,W]
    """

    def __init__(self, render_settings: TetRenderSettings):
        super().__init__()
        self.render_settings = render_settings

    def forward(self, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                tets, face_tets, tet_faces):
        f32, i32 = dict(dtype=th.float32), dict(dtype=th.int32)
        return render_tet(verts.to(**f32), faces.to(**i32), verts_color.to(**f32), faces_opacity.to(**f32),
                          mv_mats.to(**f32).transpose(1, 2), proj_mats.to(**f32).transpose(1, 2),
                          verts_depth.to(**f32), faces_intense.to(**f32),
                          tets.to(**i32), face_tets.to(**i32), tet_faces.to(**i32), self.render_settings)
