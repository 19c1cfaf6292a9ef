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

With B > 1 views there are two partitions (`partition=` of `ShardedTriRenderer`):
  * "bands": every rank renders its band of ALL views -- one band per rank, any world size;
  * "view_bands" (2 GPUs per view at C5): the tile rows of all views, view after view, are ONE sequence that is
    cut into `world` contiguous shares of equal cost (`view_shares`).  A share is a list of (view, band) segments -- one
    segment when the cuts fall on view borders (C5 on 4 or 8 ranks with equal views), two when a share straddles a border
    (views of unequal work, world % B != 0).  A rank renders each of its segments with B = 1 tensors: it bins ONE view per
    call (a B-th of the projection / set-up / scan / scatter work every rank of a "bands" run repeats); its per-view
    gradients (verts_depth, faces_intense) are rows of the flat buffer, the other views' rows stay zero.  Any world size.
  * "auto" (the default): "view_bands" when B > 1, else "bands".
Either way there is ONE gradient all-reduce over all ranks and the same flat layout.

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


# The cost model the partitions balance, in list entries.  Fitted on 21 measured shares of C5 (ms per step of one rank's work when
# the frame is cut 1 / 2 / 4 / 8 ways as bands of all views and 4 / 8 ways as (view, band) segments,
# profiles/r03/shard_cost_model_c5.txt): ms = 1.58e-7 entries + 1.31e-8 blended pairs + 6.97e-6 tiles + 0.19 per view a call bins
# (rms error 6 %, largest 11 %).  Without the pairs (only the list ranges are known): 3.76e-7 entries + 2.45e-6 tiles + 0.21 per
# view (rms 11 %, largest 19 %) -- the constants of round 3's first cut, 13 entries per tile, came from three of those shares.
TILE_COST_ENTRIES = 13.0          # ... per tile of the rows, busy or not, when only list lengths are known
PAIR_COST_ENTRIES = 0.083         # ... per blended (pixel, face) pair (`tile_hits` of a forward), and then
TILE_COST_ENTRIES_WITH_PAIRS = 44.0
# What one (view, band) call costs before any tile is composited -- projecting the vertices, setting up, scanning and scattering
# the view's faces, unpacking the gradients -- in list entries per face (C5: 2 M faces, 0.19-0.21 ms per view and call).
SEGMENT_COST_PER_FACE = 0.27
SEGMENT_COST_PER_FACE_WITH_PAIRS = 0.7


def _row_entries(ranges, B: int, gy: int, gx: int) -> np.ndarray:
    r = np.asarray(ranges).reshape(B, gy, gx, 2).astype(np.int64)
    return (r[..., 1] - r[..., 0]).sum(axis=2).astype(np.float64)  # [B, rows]


def view_row_work_from_ranges(ranges: np.ndarray, B: int, gy: int, gx: int, tile_cost: Optional[float] = None,
                              tile_hits: Optional[np.ndarray] = None) -> np.ndarray:
    """[B, rows] work per view and tile row from the per-tile [start, end) list ranges of one full forward (and, better, its
    per-tile blended-pair counts `tile_hits`, `_C.export("tile_hits", ...)`): the cost model above, in list entries.  For the
    "view_bands" partition (`ShardedTriRenderer.set_row_work`, `view_shares`)."""
    w = _row_entries(ranges, B, gy, gx)
    if tile_hits is None:
        return w + (TILE_COST_ENTRIES if tile_cost is None else tile_cost) * gx
    pairs = np.asarray(tile_hits).reshape(B, gy, gx).astype(np.int64).sum(axis=2)
    return w + PAIR_COST_ENTRIES * pairs + (TILE_COST_ENTRIES_WITH_PAIRS if tile_cost is None else tile_cost) * gx


def row_work_from_ranges(ranges: np.ndarray, B: int, gy: int, gx: int, tile_cost: Optional[float] = None,
                         tile_hits: Optional[np.ndarray] = None) -> np.ndarray:
    """Per-tile-row work summed over the views (the "bands" partition: `balanced_bands`, `set_row_work`)."""
    return view_row_work_from_ranges(ranges, B, gy, gx, tile_cost, tile_hits).sum(axis=0)


def segment_cost(F: int, with_pairs: bool) -> float:
    """What a (view, band) call costs besides its rows, in the units of view_row_work_from_ranges."""
    return (SEGMENT_COST_PER_FACE_WITH_PAIRS if with_pairs else SEGMENT_COST_PER_FACE) * F


def view_shares(view_row_work, world: int, segment_cost: float = 0.0) -> List[List[Tuple[int, int, int]]]:
    """The "view_bands" partition: `view_row_work` [B, rows] (`view_row_work_from_ranges`) -> for every rank its share, a list
    of (view, row_begin, row_end) segments.  The rows of all views, view after view, are cut into `world` contiguous shares
    whose cost -- the work of their rows + `segment_cost` per view they touch -- is as equal as contiguous cuts allow
    (smallest feasible maximum, by bisection over a greedy fill)."""
    w = np.asarray(view_row_work, dtype=np.float64) + 1e-3
    B, rows = w.shape

    def fill(limit):
        shares, cur, cost = [], [], 0.0
        for v in range(B):
            r = 0
            while r < rows:
                if cost + (0.0 if cur and cur[-1][0] == v else segment_cost) + w[v, r] > limit and cur:
                    shares.append(cur); cur, cost = [], 0.0
                    continue
                if not cur or cur[-1][0] != v:
                    cur.append([v, r, r]); cost += segment_cost
                cur[-1][2] = r + 1; cost += w[v, r]
                r += 1
        if cur:
            shares.append(cur)
        return shares

    lo, hi = float(w.max()) + segment_cost, float(w.sum()) + B * segment_cost
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if len(fill(mid)) <= world:
            hi = mid
        else:
            lo = mid
    shares = fill(hi)
    shares += [[] for _ in range(world - len(shares))]
    return [[(int(v), int(a), int(b)) for v, a, b in sh] for sh in shares]


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


def gather_view_bands(images: Sequence[Sequence[th.Tensor]], B: int, parts: Sequence[Sequence[Tuple[int, int, int]]], rank: int,
                      group=None) -> List[th.Tensor]:
    """Like gather_bands for the "view_bands" partition: images[i] are this rank's ONE-view renders ([1, C, H, W] or
    [1, H, W]) of its i-th segment, valid in that segment's pixel rows; parts[r] = the (view, row_begin, row_end) segments
    of rank r.  ONE all-gather (a rank sends the rows of its segments back to back, padded to the tallest share);
    returns the full [B, C, H, W] / [B, H, W] images."""
    world = len(parts)
    like = None
    for seg in images:
        like = [im if im.dim() == 4 else im.unsqueeze(1) for im in seg]
        break
    if like is None:
        raise ValueError("gather_view_bands: this rank has no segment to take image shapes from; pass `like` renders")
    H, W = like[0].size(2), like[0].size(3)
    chans = [im.size(1) for im in like]
    C = sum(chans)
    px = [[(v, min(H, TILE * r0), min(H, TILE * r1)) for v, r0, r1 in sh] for sh in parts]
    hmax = max(1, max(sum(y1 - y0 for _, y0, y1 in sh) for sh in px))
    send = like[0].new_zeros((1, C, hmax, W))
    at = 0
    for seg, (_, y0, y1) in zip(images, px[rank]):
        c0 = 0
        for im, c in zip(seg, chans):
            im = im if im.dim() == 4 else im.unsqueeze(1)
            send[:, c0:c0 + c, at:at + y1 - y0] = im[:, :, y0:y1]
            c0 += c
        at += y1 - y0
    recv = like[0].new_empty((world, C, hmax, W))
    dist.all_gather_into_tensor(recv, send, group=group)
    outs = [im.new_zeros((B, c, H, W)) for im, c in zip(like, chans)]
    for r, sh in enumerate(px):
        at = 0
        for v, a, b in sh:
            c0 = 0
            for o, c in zip(outs, chans):
                o[v, :, a:b] = recv[r, c0:c0 + c, at:at + b - a]
                c0 += c
            at += b - a
    return [o if im.dim() == 4 else o.squeeze(1) for o, im in zip(outs, images[0])]


class _ShardedTriViewFn(th.autograd.Function):
    """The "view_bands" partition: this rank renders its segments (view, row_begin, row_end), one B = 1 call each."""

    @staticmethod
    def forward(ctx, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense,
                settings: TriRenderSettings, segs, group, parts, impl):
        B = mv_mats.size(0)
        geom = (verts, faces, verts_color, faces_opacity)
        H, W = settings.image_height, settings.image_width
        gy = tile_rows(H)
        if not segs:  # a rank without rows still joins the collectives: an empty band of view 0
            segs = ((0, gy, gy),)
        saved, nums, outs = [], [], []
        for v, r0, r1 in segs:
            cams = _with_inverses(mv_mats[v:v + 1], proj_mats[v:v + 1])
            vdepth, fintense = verts_depth[v:v + 1].contiguous(), faces_intense[v:v + 1].contiguous()
            out = impl.render_tris(settings.bg, *geom, *cams, vdepth, fintense, H, W, rows=(r0, r1) if r1 > r0 else (gy, gy))
            outs.append((out[1], out[2]))
            nums.append(out[0])
            saved += [*cams, vdepth, fintense, *out[3:7]]
        if parts is not None and _world(group) > 1:  # assemble: one all-gather of every rank's rows
            color, depth = gather_view_bands(outs, B, parts, dist.get_rank(group), group)
        else:  # this rank's rows of its views, in place in full-size images
            color = outs[0][0].new_zeros((B,) + tuple(outs[0][0].shape[1:]))
            depth = outs[0][1].new_zeros((B,) + tuple(outs[0][1].shape[1:]))
            for (v, r0, r1), (c, z) in zip(segs, outs):
                y0, y1 = min(H, TILE * r0), min(H, TILE * r1)
                color[v, :, y0:y1] = c[0, :, y0:y1]
                depth[v, ..., y0:y1, :] = z[0, ..., y0:y1, :]
        ctx.settings, ctx.segs, ctx.group, ctx.impl, ctx.nums, ctx.B = settings, tuple(segs), group, impl, nums, B
        ctx.save_for_backward(*geom, *saved)
        return color, depth

    @staticmethod
    def backward(ctx, grad_color, grad_depth):
        saved = ctx.saved_tensors
        geom, per = saved[:4], saved[4:]
        verts, faces = geom[0], geom[1]
        P, F, B = verts.size(0), faces.size(0), ctx.B
        gy = tile_rows(ctx.settings.image_height)
        # the all-views flat layout [3P | 3P | F | B*P | B*F]: this rank fills the shared part and its views' rows of the per-view parts
        flat = th.zeros(6 * P + F + B * (P + F), dtype=th.float32, device=verts.device)
        o = 6 * P + F
        for i, (v, r0, r1) in enumerate(ctx.segs):
            s = per[10 * i:10 * i + 10]  # mv, proj, their inverses, verts_depth, faces_intense, four scratch buffers
            g = ctx.impl.render_tris_backward(ctx.settings.bg, *geom, *s[:6], grad_color[v:v + 1].contiguous(),
                                              grad_depth[v:v + 1].contiguous(), ctx.nums[i], *s[6:10],
                                              rows=(r0, r1) if r1 > r0 else (gy, gy))
            flat[:3 * P] += g[0].reshape(-1); flat[3 * P:6 * P] += g[1].reshape(-1); flat[6 * P:o] += g[2].reshape(-1)
            flat[o + v * P:o + (v + 1) * P] += g[3].reshape(-1)
            flat[o + B * P + v * F:o + B * P + (v + 1) * F] += g[4].reshape(-1)
        if dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        g_verts, g_vcolor, g_fopacity = flat[:3 * P].view(P, 3), flat[3 * P:6 * P].view(P, 3), flat[6 * P:o]
        g_vdepth, g_fintense = flat[o:o + B * P].view(B, P), flat[o + B * P:].view(B, F)
        return (g_verts, None, g_vcolor, g_fopacity, None, None, g_vdepth, g_fintense) + (None,) * 5


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
    """TriRenderer whose image is sharded across the ranks of `group`: by tile-row bands of all views ("bands") or, with
    several views, by (view, band) pairs ("view_bands"; see the module docstring).

    Same call signature as TriRenderer.  Every rank must pass identical inputs; every rank gets the
    full summed gradients.  The bands may be refreshed at any time with `set_row_work` (e.g. every few
    iterations from `row_work_from_ranges` / `view_row_work_from_ranges`)."""

    def __init__(self, render_settings: TriRenderSettings, group=None, assemble: bool = True, impl=None, partition: str = "auto"):
        super().__init__()
        if partition not in ("auto", "bands", "view_bands"):
            raise ValueError("partition must be 'auto', 'bands' or 'view_bands'")
        self.render_settings = render_settings
        self.group = group
        self.assemble = assemble
        self.partition = partition
        if impl is None:
            from . import _C as impl  # the HIP extension; fails loudly if it is not built
        self.impl = impl
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bands = equal_bands(tile_rows(render_settings.image_height), self.world)
        self.view_work = None   # "view_bands": [B, rows] work of the (view, tile row) sequence; None: every row counts the same
        self.segment_cost_per_face = SEGMENT_COST_PER_FACE  # (SEGMENT_COST_PER_FACE_WITH_PAIRS when the row work counts blended pairs)
        self._parts_key, self._parts = None, None

    def set_row_work(self, row_work, with_pairs: bool = False) -> None:
        """row_work: per tile row (summed over the views; `row_work_from_ranges`) or per view and tile row ([B, rows];
        `view_row_work_from_ranges`) -- the latter also balances the shares of the "view_bands" partition.  with_pairs: the
        work was computed with `tile_hits` (the blended pairs are in it): a (view, band) call's fixed cost is in those units."""
        w = np.asarray(row_work, dtype=np.float64)
        self.segment_cost_per_face = SEGMENT_COST_PER_FACE_WITH_PAIRS if with_pairs else SEGMENT_COST_PER_FACE
        if w.ndim == 2:
            self.view_work, self._parts_key = w.copy(), None
            w = w.sum(axis=0)
        self.bands = balanced_bands(w, self.world)

    def _use_view_bands(self, B: int) -> bool:
        return self.world > 1 and B > 1 and self.partition != "bands"

    def view_parts(self, B: int, F: int) -> List[List[Tuple[int, int, int]]]:
        """Every rank's (view, row_begin, row_end) segments under "view_bands" for B views of F faces."""
        gy = tile_rows(self.render_settings.image_height)
        key = (B, F, gy, self.world, id(self.view_work))
        if key != self._parts_key:
            if self.view_work is not None and self.view_work.shape == (B, gy):
                self._parts = view_shares(self.view_work, self.world, self.segment_cost_per_face * F)
            else:  # no statistics yet: rows of equal weight, cuts where they fall
                self._parts = view_shares(np.ones((B, gy)), self.world, 0.0)
            self._parts_key = key
        return self._parts

    @staticmethod
    def _rows(band, gy) -> Tuple[int, int]:
        r0, r1 = band
        return (gy, gy) if r1 <= r0 else (int(r0), int(r1))  # an empty band; (0, 0) would mean "all rows" to the C ABI

    @property
    def rows(self) -> Tuple[int, int]:
        return self._rows(self.bands[self.rank], tile_rows(self.render_settings.image_height))

    def forward(self, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, verts_depth, faces_intense):
        B = mv_mats.size(0)
        gy = tile_rows(self.render_settings.image_height)
        if self._use_view_bands(B):
            parts = self.view_parts(B, faces.size(0))
            return _ShardedTriViewFn.apply(verts, faces.to(dtype=th.int32), verts_color, faces_opacity,
                                           mv_mats.transpose(1, 2), proj_mats.transpose(1, 2), verts_depth, faces_intense,
                                           self.render_settings, tuple(parts[self.rank]), self.group,
                                           tuple(tuple(p) for p in parts) if self.assemble else None, self.impl)
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
