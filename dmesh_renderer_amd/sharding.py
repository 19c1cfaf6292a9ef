"""Multi-GPU sharding of the hot path: tile-row bands + one gradient all-reduce.

The reference is single-device (no collectives anywhere).  Tiles are independent in forward and
backward; the only cross-tile coupling is the SUM of gradient contributions per vertex / face.
So each rank (one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI):

  * holds the full (small) geometry and runs the binning stages with face rects clipped to its
    band of tile rows (`rows=(begin, end)` of `_C.render_tris`),
  * composites and back-propagates only its band,
  * joins ONE all-reduce(sum) over a single flattened fp32 buffer
        [dL_dverts 3P | dL_dvcolor 3P | dL_dfopacity F | dL_dvdepth B*P | dL_dfintense B*F]
    (one collective, not five: xGMI is point-to-point, small messages are latency-bound).

Forward images stay sharded unless `assemble=True`, in which case the band images (zero outside
the band) are summed with one more all-reduce so every rank sees the full image.

`impl` is the `_C`-like module used for the kernels; the default is the HIP extension.  (Tests inject
an oracle-backed stand-in to exercise this file's logic on CPU with the gloo backend.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch as th
import torch.distributed as dist

from . import TriRenderSettings, _with_inverses

TILE = 16


def tile_rows(image_height: int) -> int:
    return (image_height + TILE - 1) // TILE


def equal_bands(n_rows: int, world: int) -> List[Tuple[int, int]]:
    """First cut: contiguous bands with equal row counts."""
    cuts = [round(n_rows * k / world) for k in range(world + 1)]
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def balanced_bands(row_work: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """Contiguous bands of (nearly) equal work: split points from the prefix sum of per-row work
    (sum of tile-list lengths of the row).  Empty rows still cost a launch, hence the epsilon."""
    w = np.asarray(row_work, dtype=np.float64) + 1e-3
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, world):
        cuts.append(int(np.searchsorted(cum, cum[-1] * k / world)))
    cuts.append(len(w))
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def row_work_from_ranges(ranges: np.ndarray, B: int, gy: int, gx: int) -> np.ndarray:
    """Per-tile-row work from the per-tile [start, end) list ranges of one full forward."""
    r = np.asarray(ranges).reshape(B, gy, gx, 2).astype(np.int64)
    return (r[..., 1] - r[..., 0]).sum(axis=(0, 2))


def flatten_grads(grads: Sequence[th.Tensor], out: Optional[th.Tensor] = None) -> th.Tensor:
    flat = [g.reshape(-1) for g in grads]
    return th.cat(flat, out=out) if out is not None else th.cat(flat)


def unflatten_grads(flat: th.Tensor, like: Sequence[th.Tensor]) -> List[th.Tensor]:
    out, o = [], 0
    for g in like:
        n = g.numel()
        out.append(flat[o:o + n].view_as(g))
        o += n
    return out


def allreduce_grads(grads: Sequence[th.Tensor], group=None) -> List[th.Tensor]:
    """ONE all-reduce(sum) over the flattened gradient buffer."""
    flat = flatten_grads(grads)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return unflatten_grads(flat, grads)


class _ShardedTriFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                settings: TriRenderSettings, rows, group, assemble, impl):
        cams = _with_inverses(mv_mats, proj_mats)
        geom = (verts, faces, verts_color, faces_opacity)
        out = impl.render_tris(settings.bg, *geom, *cams, verts_depth, faces_intense,
                               settings.image_height, settings.image_width, rows=rows)
        color, depth = out[1], out[2]
        if assemble and dist.is_initialized() and dist.get_world_size(group) > 1:
            both = th.cat([color.reshape(-1), depth.reshape(-1)])
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)  # bands are zero outside their rows
            color = both[:color.numel()].view_as(color)
            depth = both[color.numel():].view_as(depth)
        ctx.settings, ctx.rows, ctx.group, ctx.impl, ctx.num_rendered = settings, rows, group, impl, out[0]
        ctx.save_for_backward(*geom, *cams, verts_depth, faces_intense, *out[3:7])
        return color, depth

    @staticmethod
    def backward(ctx, grad_color, grad_depth):
        saved = ctx.saved_tensors
        if getattr(ctx.impl, "SUPPORTS_FLAT_OUT", False):
            # the five gradients land back to back in one buffer: the all-reduce payload, no concatenation
            verts, faces, mv = saved[0], saved[1], saved[4]
            P, F, B = verts.size(0), faces.size(0), mv.size(0)
            flat = th.empty(6 * P + F + B * (P + F), dtype=th.float32, device=verts.device)
            g = ctx.impl.render_tris_backward(ctx.settings.bg, *saved[:10], grad_color, grad_depth, ctx.num_rendered,
                                              *saved[10:14], rows=ctx.rows, flat_out=flat)
            if dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
            g_verts, g_vcolor, g_fopacity, g_vdepth, g_fintense = g
        else:
            g = ctx.impl.render_tris_backward(ctx.settings.bg, *saved[:10], grad_color, grad_depth, ctx.num_rendered,
                                              *saved[10:14], rows=ctx.rows)
            g_verts, g_vcolor, g_fopacity, g_vdepth, g_fintense = allreduce_grads(g, ctx.group)
        return (g_verts, None, g_vcolor, g_fopacity, None, None, g_vdepth, g_fintense) + (None,) * 5


class ShardedTriRenderer(th.nn.Module):
    """TriRenderer whose image is sharded by tile-row bands across the ranks of `group`.

    Same call signature as TriRenderer.  Every rank must pass identical inputs; every rank gets the
    full summed gradients.  `bands` may be refreshed at any time with `set_row_work` (e.g. every few
    iterations from `row_work_from_ranges`)."""

    def __init__(self, render_settings: TriRenderSettings, group=None, assemble: bool = True, impl=None):
        super().__init__()
        self.render_settings = render_settings
        self.group = group
        self.assemble = assemble
        if impl is None:
            from . import _C as impl  # the HIP extension; fails loudly if it is not built
        self.impl = impl
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bands = equal_bands(tile_rows(render_settings.image_height), self.world)

    def set_row_work(self, row_work: Sequence[float]) -> None:
        self.bands = balanced_bands(row_work, self.world)

    @property
    def rows(self) -> Tuple[int, int]:
        r0, r1 = self.bands[self.rank]
        if r1 <= r0:  # empty band; (0, 0) would mean "all rows" to the C ABI
            gy = tile_rows(self.render_settings.image_height)
            return (gy, gy)
        return (int(r0), int(r1))

    def forward(self, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense):
        rows = self.rows if self.world > 1 else (0, 0)
        return _ShardedTriFn.apply(verts, faces.to(dtype=th.int32), verts_color, faces_opacity,
                                   mv_mats.transpose(1, 2), proj_mats.transpose(1, 2), verts_depth, faces_intense,
                                   self.render_settings, rows, self.group, self.assemble, self.impl)
