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

Forward images stay sharded unless `assemble=True`; then the bands are exchanged with ONE all-gather (every rank
sends only its own rows, padded to the tallest band: (N-1)/N of the image arrives per rank and nothing is summed;
the first version all-reduced full-size images, twice the bytes plus the adds) so every rank sees the full image.

The tet renderer shards identically (`ShardedTetRenderer`): per-pixel independence, the collective carries
[dL_dverts_color 3P | dL_dfaces_opacity F].

With B > 1 views every rank renders its band of ALL views (SURVEY 8(e) suggests (view, band) pairs: 2 GPUs per view
at C5).  Bands of all views keep one gradient collective over all ranks and one band per rank, and the per-rank
work is the same Sum over views of the band's tile lists; the per-view gradients (verts_depth, faces_intense) ride in
the same flat buffer.

`impl` is the `_C`-like module used for the kernels; the default is the HIP extension.  (Tests inject
an oracle-backed stand-in to exercise this file's logic on CPU with the gloo backend.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch as th
import torch.distributed as dist

from . import TetRenderSettings, TriRenderSettings, _with_inverses

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


# What a tile costs besides its list entries, in list entries (its workgroups in the three tile kernels, its pixels' state and
# images, its share of the scans).  From the per-rank kernel sums of C5 cut eight ways by list entries alone
# (profiles/r03/shard_kernel_sums_c5.json, first cut: the edge bands, 72 tile rows, took 2.14 ms, the middle band, 19 rows with
# the same number of entries, 1.64 ms): 0.5 ms per 54 000 tiles against 1.64 ms per 2.34 M entries.
TILE_COST_ENTRIES = 13.0


def row_work_from_ranges(ranges: np.ndarray, B: int, gy: int, gx: int, tile_cost: float = TILE_COST_ENTRIES) -> np.ndarray:
    """Per-tile-row work from the per-tile [start, end) list ranges of one full forward: list entries + a cost per tile."""
    r = np.asarray(ranges).reshape(B, gy, gx, 2).astype(np.int64)
    return (r[..., 1] - r[..., 0]).sum(axis=(0, 2)) + tile_cost * B * gx


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


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def gather_bands(images: Sequence[th.Tensor], bands: Sequence[Tuple[int, int]], rank: int, group=None) -> List[th.Tensor]:
    """Assemble full images from tile-row bands with ONE all-gather.  `images` are this rank's [B, C, H, W] (or
    [B, H, W]) renders, valid in the pixel rows of bands[rank]; all of them travel in one buffer.  Every rank
    contributes rows_max * 16 pixel rows (its band, zero padded), so the collective is a plain equal-size
    all_gather_into_tensor; the receiver copies each rank's rows into place."""
    world = len(bands)
    ims = [im if im.dim() == 4 else im.unsqueeze(1) for im in images]
    B, H, W = ims[0].size(0), ims[0].size(2), ims[0].size(3)
    chans = [im.size(1) for im in ims]
    C = sum(chans)
    px = [(min(H, TILE * r0), min(H, TILE * r1)) for r0, r1 in bands]
    hmax = max(1, max(y1 - y0 for y0, y1 in px))
    send = ims[0].new_zeros((B, C, hmax, W))
    y0, y1 = px[rank]
    if y1 > y0:
        c0 = 0
        for im, c in zip(ims, chans):
            send[:, c0:c0 + c, :y1 - y0] = im[:, :, y0:y1]
            c0 += c
    recv = ims[0].new_empty((world * B, C, hmax, W))  # concatenation along dim 0 (what gloo's all_gather_into_tensor accepts)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, B, C, hmax, W)
    outs = [th.empty_like(im) for im in ims]
    for k, (a, b) in enumerate(px):
        if b <= a:
            continue
        c0 = 0
        for o, c in zip(outs, chans):
            o[:, :, a:b] = recv[k, :, c0:c0 + c, :b - a]
            c0 += c
    return [o if im.dim() == 4 else o.squeeze(1) for o, im in zip(outs, images)]


class _ShardedTriFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                settings: TriRenderSettings, rows, group, bands, impl):
        cams = _with_inverses(mv_mats, proj_mats)
        geom = (verts, faces, verts_color, faces_opacity)
        out = impl.render_tris(settings.bg, *geom, *cams, verts_depth, faces_intense,
                               settings.image_height, settings.image_width, rows=rows)
        color, depth = out[1], out[2]
        if bands is not None and _world(group) > 1:  # assemble: one all-gather of the bands
            color, depth = gather_bands((color, depth), bands, dist.get_rank(group), group)
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
        bands = tuple(self.bands) if self.assemble and self.world > 1 else None
        return _ShardedTriFn.apply(verts, faces.to(dtype=th.int32), verts_color, faces_opacity,
                                   mv_mats.transpose(1, 2), proj_mats.transpose(1, 2), verts_depth, faces_intense,
                                   self.render_settings, rows, self.group, bands, self.impl)


class _ShardedTetFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                tets, face_tets, tet_faces, settings: TetRenderSettings, rows, group, bands, impl):
        cams = _with_inverses(mv_mats, proj_mats)
        geom = (verts, faces, verts_color, faces_opacity)
        topo = (tets, face_tets, tet_faces)
        out = impl.render_tets(settings.bg, *geom, *cams, verts_depth, faces_intense, *topo,
                               settings.image_height, settings.image_width, settings.ray_random_seed, rows=rows)
        color, depth, active = out[0], out[1], out[2]
        if bands is not None and _world(group) > 1:
            color, depth, active = gather_bands((color, depth, active), bands, dist.get_rank(group), group)
        active = active > 0.5  # bool mask, reference __init__.py:333
        ctx.settings, ctx.rows, ctx.group, ctx.impl = settings, rows, group, impl
        ctx.save_for_backward(*geom, *cams, verts_depth, faces_intense, *topo, *out[3:7])
        ctx.mark_non_differentiable(active)
        return color, depth, active

    @staticmethod
    def backward(ctx, grad_color, grad_depth, _grad_active):
        saved = ctx.saved_tensors
        verts, faces = saved[0], saved[1]
        P, F = verts.size(0), faces.size(0)
        if getattr(ctx.impl, "SUPPORTS_FLAT_OUT", False):
            flat = th.empty(3 * P + F, dtype=th.float32, device=verts.device)  # [dL_dverts_color 3P | dL_dfaces_opacity F]
            g_vcolor, g_fopacity = ctx.impl.render_tets_backward(ctx.settings.bg, *saved[:13], grad_color, grad_depth,
                                                                 *saved[13:17], rows=ctx.rows, flat_out=flat)
            if _world(ctx.group) > 1:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        else:
            g = ctx.impl.render_tets_backward(ctx.settings.bg, *saved[:13], grad_color, grad_depth, *saved[13:17],
                                              rows=ctx.rows)
            g_vcolor, g_fopacity = allreduce_grads(g, ctx.group)
        return (None, None, g_vcolor, g_fopacity) + (None,) * 12


class ShardedTetRenderer(th.nn.Module):
    """TetRenderer whose image is sharded by tile-row bands across the ranks of `group` ("the tet path
    shards identically"): every rank bins with rects clipped to its band, finds first hits and marches the rays of
    its band only, back-propagates them, and joins ONE all-reduce over [dL_dverts_color 3P | dL_dfaces_opacity F].
    Same call signature and outputs as TetRenderer (color, depth, active bool)."""

    def __init__(self, render_settings: TetRenderSettings, group=None, assemble: bool = True, impl=None):
        super().__init__()
        self.render_settings = render_settings
        self.group = group
        self.assemble = assemble
        if impl is None:
            from . import _C as impl  # the HIP extension; fails loudly if it is not built
        self.impl = impl
        self.world = _world(group)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bands = equal_bands(tile_rows(render_settings.image_height), self.world)

    def set_row_work(self, row_work: Sequence[float]) -> None:
        self.bands = balanced_bands(row_work, self.world)

    @property
    def rows(self) -> Tuple[int, int]:
        r0, r1 = self.bands[self.rank]
        if r1 <= r0:
            gy = tile_rows(self.render_settings.image_height)
            return (gy, gy)
        return (int(r0), int(r1))

    def forward(self, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                tets, face_tets, tet_faces):
        f32, i32 = dict(dtype=th.float32), dict(dtype=th.int32)
        rows = self.rows if self.world > 1 else (0, 0)
        bands = tuple(self.bands) if self.assemble and self.world > 1 else None
        return _ShardedTetFn.apply(verts.to(**f32), faces.to(**i32), verts_color.to(**f32), faces_opacity.to(**f32),
                                   mv_mats.to(**f32).transpose(1, 2), proj_mats.to(**f32).transpose(1, 2),
                                   verts_depth.to(**f32), faces_intense.to(**f32),
                                   tets.to(**i32), face_tets.to(**i32), tet_faces.to(**i32),
                                   self.render_settings, rows, self.group, bands, self.impl)
