"""tri renderer: HIP path (through `_C` -> C ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): tile/sort indices bit-exact; forward pixels <= 1e-5;
gradients <= 1e-4 (max-abs error relative to max(1, max-abs of the oracle gradient)).
"""
import numpy as np
import pytest
import torch as th

from dmesh_renderer_amd import scenes
from util import SUM_ORDER_TOL, c_args, rel_err, sum_order_tol, upstream_grads

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5
GRAD_TOL = 1e-4

CASES = {
    # name: (layers, n, B, H, W, opacity)
    "C1": (4, 17, 1, 256, 256, (0.1, 0.5)),                 # BASELINE configs[0]
    "C1_opaque": (4, 17, 1, 256, 256, (0.5, 0.95)),         # exercises early termination (T < 1e-4)
    "alpha_one": (4, 17, 1, 256, 256, (0.2, 0.6)),          # every 5th face has opacity exactly 1 (Q10's alpha == 1 branch)
    "ragged": (3, 12, 2, 200, 328, (0.1, 0.6)),             # W, H not multiples of 16; two views
    "dense": (12, 9, 1, 64, 64, (0.05, 0.3)),               # long tile lists: several LDS chunks per tile
    # 5120 faces over 4 tiles: lists longer than the sort's LDS capacity (in-place bitonic network in HBM), 40 chunks
    "very_dense": (40, 9, 1, 32, 32, (0.01, 0.08)),
    # 2 x 96 x 96 = 18 432 tiles: the multi-workgroup tile scans (more than 8 192 tiles) and windowed binning
    "many_tiles": (3, 24, 2, 1536, 1536, (0.1, 0.5)),
    # 90 x 91 = 8 190 tiles: the single-workgroup scans with their LDS slab (8 192 tiles) all but full, and
    # 91 x 92 = 8 372 tiles: two scan workgroups, the second one nearly empty
    "slab_full": (3, 20, 1, 1440, 1456, (0.1, 0.5)),
    "two_scan_blocks": (3, 20, 1, 1456, 1472, (0.1, 0.5)),
    # 300 consecutive screen-filling triangles (400 tiles each): more than the binning kernels' LDS queue of big
    # faces holds per workgroup (128), so both the cooperative emission and its per-thread fallback run
    "many_big": (150, 2, 1, 320, 320, (0.005, 0.02)),
    # triangles far larger than the image, vertices far off-screen and behind the camera (mirrored by
    # clamp_w, Q2): exercises whole-tile coverage, the non-"near" / 32-bit coverage paths and int32 wrap (Q7)
    "huge": (3, 4, 2, 96, 144, (0.1, 0.4)),
    # four triangles spanning ~4000 px of a 4096^2 image (65 536 sub-pixel units): the int32 products of in_tri wrap
    # (Q7) for tiles far from the vertices, so the per-tile pixel box may only be trusted near them (edge_setup's
    # 2^14 bound); every one of the 65 536 tiles holds a short list
    "span_4000": (2, 2, 1, 4096, 4096, (0.1, 0.4)),
    # a small mesh in the middle of a 2048^2 frame: 16 384 tiles, far fewer list entries than tiles -- the coverage masks'
    # first-chunk slots are numbered by the tile's rank among the busy tiles and there are only min(tiles, entries) of them
    "sparse_tiles": (3, 8, 1, 2048, 2048, (0.1, 0.5)),
}


def _make(case):
    L, n, B, H, W, op = CASES[case]
    d = scenes.layered_sheets(L, n, B, H, W, seed=0, opacity=op)
    if case == "huge":
        d["verts"] = d["verts"] * th.tensor([40.0, 40.0, 3.0])
    if case == "span_4000":
        d["verts"] = d["verts"] * th.tensor([1.7, 1.7, 1.0])
    if case == "sparse_tiles":
        d["verts"] = d["verts"] * th.tensor([0.2, 0.2, 1.0])
    if case == "many_big":
        d["verts"] = d["verts"] * th.tensor([6.0, 6.0, 1.0])
    if case == "alpha_one":
        d["faces_opacity"][::5] = 1.0
    return d, B, H, W


def _run(oracle, dev, d, H, W, rows=(0, 0)):
    from dmesh_renderer_amd import _C
    sc = oracle.scene_from_module_inputs(d, H, W, rows=rows)
    ocolor, odepth, ost = oracle.tri_forward(sc)
    args = c_args(d, dev)
    out = _C.render_tris(*args, H, W, rows=rows)
    th.cuda.synchronize()
    return sc, (ocolor, odepth, ost), args, out


@pytest.mark.parametrize("case", list(CASES))
def test_forward_and_indices(oracle, hip_device, case):
    from dmesh_renderer_amd import _C
    d, B, H, W = _make(case)
    sc, (ocolor, odepth, ost), args, out = _run(oracle, hip_device, d, H, W)
    R, color, depth, bufs = out[0], out[1], out[2], out[3:7]
    assert R == ost.num_rendered

    def ex(name, dtype):
        return _C.export(name, args, False, R, bufs, H, W, dtype).cpu().numpy()

    # bit-exact integer / index work
    np.testing.assert_array_equal(ex("image", th.float32).view(np.uint32), ost.get("image").view(np.uint32))
    np.testing.assert_array_equal(ex("ndc_z", th.float32).view(np.uint32),
                                  ost.get("ndc").reshape(-1, 3)[:, 2].copy().view(np.uint32))
    np.testing.assert_array_equal(ex("tiles_touched", th.int32).view(np.uint32), ost.get("tiles_touched"))
    touched = ost.get("tiles_touched") > 0
    np.testing.assert_array_equal(ex("key_depth", th.float32).view(np.uint32)[touched],
                                  ost.get("depths").view(np.uint32)[touched])
    np.testing.assert_array_equal(ex("ranges", th.int32).view(np.uint32), ost.get("ranges"))
    np.testing.assert_array_equal(ex("face_list", th.int32).view(np.uint32), ost.get("values"))
    np.testing.assert_array_equal(ex("n_contrib", th.int32).view(np.uint32), ost.get("n_contrib"))
    # floating point outputs
    assert np.abs(color.cpu().numpy() - ocolor).max() <= FWD_TOL
    assert np.abs(depth.cpu().numpy() - odepth).max() <= FWD_TOL
    assert np.abs(ex("final_T", th.float32) - ost.get("final_T")).max() <= FWD_TOL
    if case == "C1_opaque":
        assert (ost.get("final_T") < 1e-4).any(), "scene must exercise early termination"


@pytest.mark.parametrize("case", list(CASES))
def test_backward(oracle, hip_device, case):
    from dmesh_renderer_amd import _C
    d, B, H, W = _make(case)
    sc, (ocolor, odepth, ost), args, out = _run(oracle, hip_device, d, H, W)
    gc, gd = upstream_grads(B, H, W)
    og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
    g = _C.render_tris_backward(*args, gc.to(hip_device), gd.to(hip_device), out[0], *out[3:7])
    th.cuda.synchronize()
    for got, key in zip(g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
        assert got.shape == og[key].shape
        e = rel_err(got.cpu().numpy(), og[key])
        assert e <= GRAD_TOL, f"{key}: {e}"


def test_band_rows_compose(oracle, hip_device):
    """Two tile-row bands rendered separately == the full render (forward pixels and summed grads)."""
    from dmesh_renderer_amd import _C
    L, n, B, H, W, op = CASES["ragged"]
    d = scenes.layered_sheets(L, n, B, H, W, seed=3, opacity=op)
    args = c_args(d, hip_device)
    gc, gd = upstream_grads(B, H, W)
    gc, gd = gc.to(hip_device), gd.to(hip_device)
    full = _C.render_tris(*args, H, W)
    gfull = _C.render_tris_backward(*args, gc, gd, full[0], *full[3:7])
    gy = (H + 15) // 16
    split = gy // 2
    color = th.zeros_like(full[1]); depth = th.zeros_like(full[2])
    gsum = [th.zeros_like(t) for t in gfull]
    for rows in ((0, split), (split, gy)):
        o = _C.render_tris(*args, H, W, rows=rows)
        color += o[1]; depth += o[2]
        gb = _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7], rows=rows)
        for a, b_ in zip(gsum, gb):
            a += b_
    assert th.equal(color, full[1]) and th.equal(depth, full[2])
    for a, b_ in zip(gsum, gfull):
        assert rel_err(a.cpu().numpy(), b_.cpu().numpy()) <= GRAD_TOL


def test_flat_gradient_output(hip_device):
    """flat_out: the five gradients written back to back into one caller-owned buffer (the all-reduce payload)."""
    from dmesh_renderer_amd import _C
    L, n, B, H, W, op = CASES["ragged"]
    d = scenes.layered_sheets(L, n, B, H, W, seed=5, opacity=op)
    args = c_args(d, hip_device)
    gc, gd = upstream_grads(B, H, W)
    gc, gd = gc.to(hip_device), gd.to(hip_device)
    o = _C.render_tris(*args, H, W)
    g = _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7])
    P, F = d["verts"].shape[0], d["faces"].shape[0]
    flat = th.full((6 * P + F + B * (P + F),), float("nan"), device=hip_device)
    gv = _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7], flat_out=flat)
    assert all(v.data_ptr() >= flat.data_ptr() for v in gv) and not th.isnan(flat).any()
    for a, b_, k in zip(gv, g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
        assert a.shape == b_.shape and rel_err(a.cpu().numpy(), b_.cpu().numpy()) <= sum_order_tol(k), k
    assert rel_err(flat.cpu().numpy(), th.cat([t.reshape(-1) for t in g]).cpu().numpy()) <= SUM_ORDER_TOL
    with pytest.raises(RuntimeError, match="flat_out"):
        _C.render_tris_backward(*args, gc, gd, o[0], *o[3:7], flat_out=flat[:-1])


def test_size_guess_is_refuted_and_redone(oracle, hip_device):
    """Speculative sizing: the library sizes the binning / hit-record buffers from the previous call with the
    same tensor shapes.  Render a small-on-screen mesh first, then the same shapes filling the screen (R and the
    hit count grow far beyond the +25 % guess): the stages must be redone with exact sizes, results exact."""
    from dmesh_renderer_amd import _C
    L, n, B, H, W = 3, 14, 1, 160, 160
    gc, gd = upstream_grads(B, H, W)
    rs = []
    redo0 = _C.redo_count()
    for scale in (0.12, 1.0, 0.3, 1.0):
        d = scenes.layered_sheets(L, n, B, H, W, seed=7)
        d["verts"] = d["verts"] * scale
        sc = oracle.scene_from_module_inputs(d, H, W)
        ocolor, odepth, ost = oracle.tri_forward(sc)
        og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
        args = c_args(d, hip_device)
        out = _C.render_tris(*args, H, W)
        g = _C.render_tris_backward(*args, gc.to(hip_device), gd.to(hip_device), out[0], *out[3:7])
        th.cuda.synchronize()
        rs.append(out[0])
        assert out[0] == ost.num_rendered
        assert np.abs(out[1].cpu().numpy() - ocolor).max() <= FWD_TOL
        for got, key in zip(g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, (scale, key)
    assert rs[1] > 1.5 * rs[0], "second scene must exceed the size guess"
    assert _C.redo_count() >= redo0 + 2, "the refuted guesses must have gone through the redo path (forward and backward)"


def test_size_guess_refuted_by_screen_filling_faces(oracle, hip_device):
    """Same view configuration, list-entries-per-face ratio up by 3x: a mesh, then the mesh plus screen-filling
    triangles.  The guessed binning buffer is far too small; every kernel that walks the tile lists must stay
    inside it until the stages are redone (this scenario once read and wrote past the buffer), and the
    screen-filling faces go through the cooperative emission of the binning kernels."""
    from dmesh_renderer_amd import _C
    B, H, W = 1, 320, 320
    gc, gd = upstream_grads(B, H, W)
    base = scenes.layered_sheets(4, 24, B, H, W, seed=11)
    big = scenes.layered_sheets(16, 2, B, H, W, seed=12, opacity=(0.01, 0.05))   # 32 triangles, one quad per layer
    big["verts"] = big["verts"] * th.tensor([6.0, 6.0, 1.0])
    P0 = base["verts"].shape[0]
    both = dict(base)
    both["verts"] = th.cat([base["verts"], big["verts"]]); both["verts_color"] = th.cat([base["verts_color"], big["verts_color"]])
    both["faces"] = th.cat([base["faces"], big["faces"] + P0]); both["faces_opacity"] = th.cat([base["faces_opacity"], big["faces_opacity"]])
    both["verts_depth"] = th.cat([base["verts_depth"], big["verts_depth"]], dim=1)
    both["faces_intense"] = th.cat([base["faces_intense"], big["faces_intense"]], dim=1)
    rs = []
    redo0 = _C.redo_count()
    for d in (base, both, base, both):
        sc = oracle.scene_from_module_inputs(d, H, W)
        ocolor, odepth, ost = oracle.tri_forward(sc)
        og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
        args = c_args(d, hip_device)
        out = _C.render_tris(*args, H, W)
        g = _C.render_tris_backward(*args, gc.to(hip_device), gd.to(hip_device), out[0], *out[3:7])
        th.cuda.synchronize()
        assert out[0] == ost.num_rendered
        np.testing.assert_array_equal(_C.export("face_list", args, False, out[0], out[3:7], H, W, th.int32).cpu().numpy().view(np.uint32),
                                      ost.get("values"))
        assert np.abs(out[1].cpu().numpy() - ocolor).max() <= FWD_TOL
        for got, key in zip(g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, key
        rs.append(out[0] / d["faces"].shape[0])
    assert rs[1] > 2.0 * rs[0], "the second scene must refute the size guess by a wide margin"
    assert _C.redo_count() >= redo0 + 2, "the refuted guesses must have gone through the redo path"


@pytest.mark.parametrize("scale,rows", [(4.0, (0, 0)), (0.45, (0, 0)), (4.0, (9, 41))])
def test_record_regions_without_a_scan_launch(oracle, hip_device, scale, rows):
    """A frame of exactly 8 192 tiles (2048 x 1024), the most k_tri_backward_pix lays the record regions out for by itself
    (dmr_kernels.hpp, HitRegions).  The first backward of a view configuration has no size estimate and goes through
    k_scan_hits; every later one sums the forward's per-tile record bounds inside the per-pixel kernel, the last tile's
    workgroup leaves the total -- with the last tile busy (scale 4: the mesh covers the frame), empty (scale 0.45: only its
    total is wanted of that workgroup) and outside the rendered band of tile rows.  Both must give the oracle's gradients."""
    from dmesh_renderer_amd import _C
    L, n, B, H, W = 3, 40, 1, 1024, 2048
    assert ((H + 15) // 16) * ((W + 15) // 16) * B == 8192
    d = scenes.layered_sheets(L, n, B, H, W, seed=5, opacity=(0.1, 0.5))
    d["verts"] = d["verts"] * th.tensor([scale, scale, 1.0])
    sc = oracle.scene_from_module_inputs(d, H, W, rows=rows)
    ocolor, odepth, ost = oracle.tri_forward(sc)
    gc, gd = upstream_grads(B, H, W)
    og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
    args = c_args(d, hip_device)
    last_tile_busy = None
    for call in range(3):  # scan kernel, then twice the per-pixel kernel's own layout (default call, asynchronous call)
        _C.set_async(call == 2)
        try:
            out = _C.render_tris(*args, H, W, rows=rows)
            g = _C.render_tris_backward(*args, gc.to(hip_device), gd.to(hip_device), out[0], *out[3:7], rows=rows)
            th.cuda.synchronize()
        finally:
            _C.set_async(False)
        assert not _C.overflowed()
        if call < 2:
            assert out[0] == ost.num_rendered
        hits = _C.export("tile_hits", args, False, out[0], out[3:7], H, W, th.int32).cpu().numpy()
        last_tile_busy = bool(hits[-1] > 0)
        assert np.abs(out[1].cpu().numpy() - ocolor).max() <= FWD_TOL
        for got, key in zip(g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, (call, key)
    assert last_tile_busy == (scale > 1.0 and rows == (0, 0))


def test_module_autograd(oracle, hip_device):
    """TriRenderer Module: loss.backward() routes the five gradients like the reference wrapper."""
    import dmesh_renderer_amd as dmr
    L, n, B, H, W, op = CASES["C1"]
    d = scenes.layered_sheets(L, n, B, H, W, seed=0, opacity=op)
    dev = hip_device
    t = {k: v.to(dev) for k, v in d.items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")}
    r = dmr.TriRenderer(dmr.TriRenderSettings(H, W, t["bg"]))
    color, depth = r(leaves["verts"], t["faces"].to(th.int64), leaves["verts_color"], leaves["faces_opacity"],
                     t["mv_mats"], t["proj_mats"], leaves["verts_depth"], leaves["faces_intense"])
    gc, gd = upstream_grads(B, H, W)
    ((color * gc.to(dev)).sum() + (depth * gd.to(dev)).sum()).backward()
    sc = oracle.scene_from_module_inputs(d, H, W)
    ocolor, odepth, ost = oracle.tri_forward(sc)
    og = oracle.tri_backward(sc, ost, gc.numpy(), gd.numpy())
    assert np.abs(color.detach().cpu().numpy() - ocolor).max() <= FWD_TOL
    for k in leaves:
        assert rel_err(leaves[k].grad.cpu().numpy(), og[k]) <= GRAD_TOL, k


def test_empty_inputs(hip_device):
    """P == 0 / F == 0: zeros and num_rendered == 0 without launching (render.cu:104-105,173; Q16)."""
    from dmesh_renderer_amd import _C
    dev = hip_device
    d = scenes.layered_sheets(1, 3, 1, 32, 32)
    for P, F in ((0, 0), (9, 0)):
        dd = dict(d)
        dd["verts"] = d["verts"][:P]; dd["verts_color"] = d["verts_color"][:P]; dd["verts_depth"] = d["verts_depth"][:, :P]
        dd["faces"] = d["faces"][:F]; dd["faces_opacity"] = d["faces_opacity"][:F]; dd["faces_intense"] = d["faces_intense"][:, :F]
        args = c_args(dd, dev)
        out = _C.render_tris(*args, 32, 32)
        assert out[0] == 0 and float(out[1].abs().max()) == 0.0 and float(out[2].abs().max()) == 0.0
        g = _C.render_tris_backward(*args, th.ones(1, 3, 32, 32, device=dev), th.ones(1, 1, 32, 32, device=dev), out[0], *out[3:7])
        assert [tuple(x.shape) for x in g] == [(P, 3), (P, 3), (F,), (1, P), (1, F)]
        assert all(float(x.abs().sum()) == 0.0 for x in g)


def test_shape_errors(hip_device):
    from dmesh_renderer_amd import _C
    d = scenes.layered_sheets(1, 3, 1, 32, 32)
    args = c_args(d, hip_device)
    bad = list(args); bad[1] = bad[1][:, :2]
    with pytest.raises(RuntimeError, match="verts must have dimensions"):
        _C.render_tris(*bad, 32, 32)
    bad = list(args); bad[4] = bad[4][:-1]
    with pytest.raises(RuntimeError, match="face opacity must have dimensions"):
        _C.render_tris(*bad, 32, 32)
    bad = list(args); bad[2] = bad[2].to(th.int64)
    with pytest.raises(RuntimeError, match="expected scalar type Int"):
        _C.render_tris(*bad, 32, 32)
    cpu = c_args(d, None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.render_tris(*cpu, 32, 32)


def test_empty_band_renders_nothing(oracle, hip_device):
    """A rank whose band is empty (balanced_bands can hand one out when a single tile row holds more than 1/world of
    the work): rows=(r, r) with r > 0 renders nothing -- zero image, zero gradients, num_rendered 0 -- and a band
    clamped away beyond the last row behaves the same."""
    from dmesh_renderer_amd import _C
    d, B, H, W = _make("ragged")
    args = c_args(d, hip_device)
    gc, gd = upstream_grads(B, H, W)
    gy = (H + 15) // 16
    for rows in ((3, 3), (gy, gy + 4)):
        out = _C.render_tris(*args, H, W, rows=rows)
        th.cuda.synchronize()
        assert out[0] == 0
        assert float(out[1].abs().max()) == 0.0 and float(out[2].abs().max()) == 0.0
        g = _C.render_tris_backward(*args, gc.to(hip_device), gd.to(hip_device), out[0], *out[3:7], rows=rows)
        th.cuda.synchronize()
        for t in g:
            assert float(t.abs().max()) == 0.0


@pytest.mark.parametrize("copies,layers", [(40, 3), (150, 2), (700, 1), (20, 8)])
def test_sort_with_equal_and_bunched_depths(oracle, hip_device, copies, layers):
    """The tile sort's bucket path (dmr_sort.hpp): every face of a small layered scene duplicated `copies` times -- the copies
    have bit-identical depth keys and sort by face id -- so a tile's list is `layers` bunches of equal keys: bunches that fit a
    bucket are ranked inside it (40, 20 copies), larger ones send the tile to the segmented rank sort (150), lists beyond 2 048
    entries to the bitonic network (700).  face_list must be the reference's stable (depth, emission order) order bit for bit."""
    from dmesh_renderer_amd import _C
    B, H, W = 1, 48, 64
    d = scenes.layered_sheets(layers, 3, B, H, W, seed=7, opacity=(0.002, 0.01))
    F0 = d["faces"].shape[0]
    d["faces"] = d["faces"].repeat(copies, 1).contiguous()
    d["faces_opacity"] = d["faces_opacity"].repeat(copies).contiguous()
    d["faces_intense"] = d["faces_intense"].repeat(1, copies).contiguous()
    assert d["faces"].shape[0] == F0 * copies
    sc, (ocolor, odepth, ost), args, out = _run(oracle, hip_device, d, H, W)
    R, bufs = out[0], out[3:7]
    assert R == ost.num_rendered
    ex = lambda name, dtype: _C.export(name, args, False, R, bufs, H, W, dtype).cpu().numpy()
    ranges = ost.get("ranges").reshape(-1, 2)
    longest = int((ranges[:, 1].astype(np.int64) - ranges[:, 0].astype(np.int64)).max())
    assert longest > 64 and (copies != 700 or longest > 2048), longest  # the lists this test is about
    np.testing.assert_array_equal(ex("ranges", th.int32).view(np.uint32), ost.get("ranges"))
    np.testing.assert_array_equal(ex("face_list", th.int32).view(np.uint32), ost.get("values"))
    np.testing.assert_array_equal(ex("n_contrib", th.int32).view(np.uint32), ost.get("n_contrib"))
    assert np.abs(out[1].cpu().numpy() - ocolor).max() <= FWD_TOL
