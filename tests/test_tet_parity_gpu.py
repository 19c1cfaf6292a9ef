"""tet renderer: HIP path (through `_C` -> C ABI) vs the CPU oracle on the same seeded inputs.

expf/logf differ by ulps between glibc and the device library, so floating outputs are
tolerance-checked (forward 1e-5, gradients 1e-4); binning indices and the per-pixel march
topology (first/last face and tet, n_contrib, active) are compared exactly.
"""
import numpy as np
import pytest
import torch as th

from dmesh_renderer_amd import scenes
from util import c_args, rel_err, upstream_grads

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5
GRAD_TOL = 1e-4

CASES = {
    # name: (m, B, H, W, opacity)
    "small": (4, 1, 128, 128, (0.02, 0.3)),
    "two_views_ragged": (5, 2, 120, 200, (0.05, 0.5)),   # H, W not multiples of 16 (Q18 guarded)
    "opaque": (6, 1, 96, 96, (0.6, 1.0)),                # early termination; opacity == 1 is clamped below
}


def _scene(case):
    m, B, H, W, op = CASES[case]
    d = scenes.kuhn_tets(m, B, H, W, seed=0, opacity=op)
    if case == "opaque":
        d["faces_opacity"][::7] = 1.0  # exercises the alpha == 1 branches
    return d, B, H, W


@pytest.mark.parametrize("case", list(CASES))
def test_forward_and_topology(oracle, hip_device, case):
    from dmesh_renderer_amd import _C
    d, B, H, W = _scene(case)
    sc = oracle.scene_from_module_inputs(d, H, W)
    ocolor, odepth, oactive, ost = oracle.tet_forward(sc)
    args = c_args(d, hip_device, tet=True)
    out = _C.render_tets(*args, H, W, 0)
    th.cuda.synchronize()
    color, depth, active, bufs = out[0], out[1], out[2], out[3:7]
    R = ost.num_rendered

    def ex(name, dtype):
        return _C.export(name, args, True, R, bufs, H, W, dtype).cpu().numpy()

    np.testing.assert_array_equal(ex("tiles_touched", th.int32).view(np.uint32), ost.get("tiles_touched"))
    touched = ost.get("tiles_touched") > 0
    np.testing.assert_array_equal(ex("key_depth", th.float32).view(np.uint32)[touched], ost.get("min_depths").view(np.uint32)[touched])
    np.testing.assert_array_equal(ex("max_depth", th.float32).view(np.uint32)[touched], ost.get("max_depths").view(np.uint32)[touched])
    np.testing.assert_array_equal(ex("ranges", th.int32).view(np.uint32), ost.get("ranges"))
    np.testing.assert_array_equal(ex("face_list", th.int32).view(np.uint32), ost.get("values"))
    for name in ("first_face", "first_tet", "last_face", "last_tet"):
        np.testing.assert_array_equal(ex(name, th.int32), ost.get(name), err_msg=name)
    np.testing.assert_array_equal(ex("n_contrib", th.int32).view(np.uint32), ost.get("n_contrib"))
    np.testing.assert_array_equal(active.cpu().numpy(), oactive)
    assert oactive.mean() > 0.2, "scene must have marched pixels"
    assert np.abs(color.cpu().numpy() - ocolor).max() <= FWD_TOL
    assert np.abs(depth.cpu().numpy() - odepth).max() <= FWD_TOL


@pytest.mark.parametrize("case", list(CASES))
def test_backward(oracle, hip_device, case):
    from dmesh_renderer_amd import _C
    d, B, H, W = _scene(case)
    sc = oracle.scene_from_module_inputs(d, H, W)
    _, _, _, ost = oracle.tet_forward(sc)
    gc, gd = upstream_grads(B, H, W)
    og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
    args = c_args(d, hip_device, tet=True)
    out = _C.render_tets(*args, H, W, 0)
    g = _C.render_tets_backward(*args, gc.to(hip_device), gd.to(hip_device), *out[3:7])
    th.cuda.synchronize()
    for got, key in zip(g, ("verts_color", "faces_opacity")):
        e = rel_err(got.cpu().numpy(), og[key])
        assert e <= GRAD_TOL, f"{key}: {e}"


def _seq_state(_C, args, bufs, H, W):
    """(longest march of the forward in steps, steps per pixel its march sequence had room for)"""
    longest, cap = _C.export("tet_seq", args, True, 0, bufs, H, W, th.int32).cpu().numpy().view(np.uint32)[:2]
    return int(longest), int(cap)


@pytest.mark.parametrize("case", list(CASES))
def test_backward_on_the_march_sequence(oracle, hip_device, case):
    """The backward of the SECOND call of a view configuration consumes the faces the forward marched through (its
    sequence fits the capacity the first call's backward reported) instead of re-marching like the reference
    (cuda_renderer/backward.cu:372-476): same gradients, and they repeat bit for bit from one forward."""
    from dmesh_renderer_amd import _C
    d, B, H, W = _scene(case)
    W = W + 16 * (1 + list(CASES).index(case))  # a view configuration no other test uses: the first call has no estimate
    d = scenes.kuhn_tets(CASES[case][0], B, H, W, seed=0, opacity=CASES[case][4])
    if case == "opaque":
        d["faces_opacity"][::7] = 1.0
    sc = oracle.scene_from_module_inputs(d, H, W)
    _, _, _, ost = oracle.tet_forward(sc)
    gc, gd = upstream_grads(B, H, W)
    og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
    args = c_args(d, hip_device, tet=True)
    grads = []
    for call in range(3):
        out = _C.render_tets(*args, H, W, 0)
        g = _C.render_tets_backward(*args, gc.to(hip_device), gd.to(hip_device), *out[3:7])
        th.cuda.synchronize()
        longest, cap = _seq_state(_C, args, out[3:7], H, W)
        assert longest == int(ost.get("n_contrib").max())
        if call == 0:
            assert cap == 0, "first call of a view configuration: no estimate, the backward re-marches"
        else:
            assert 0 < longest <= cap, (longest, cap)  # the sequence is complete: k_tet_backward_seq did the work
        for got, key in zip(g, ("verts_color", "faces_opacity")):
            e = rel_err(got.cpu().numpy(), og[key])
            assert e <= GRAD_TOL, f"call {call} {key}: {e}"
        grads.append([x.cpu().numpy() for x in g])
    # sequence path vs re-march path: the same (pixel, face) values, summed in a different order
    for a, b_ in zip(grads[0], grads[1]):
        assert rel_err(a, b_) <= 1e-5
    # the topology the forward reports is unchanged by the sequence stores
    np.testing.assert_array_equal(_C.export("n_contrib", args, True, 0, out[3:7], H, W, th.int32).cpu().numpy().view(np.uint32), ost.get("n_contrib"))


def test_march_sequence_overflow_falls_back(oracle, hip_device):
    """A step whose marches outgrow the sequence capacity estimated from the previous call (the same mesh and view configuration,
    first nearly opaque -- every ray stops after a few faces --, then nearly transparent): the forward's sequence is incomplete,
    the device-side check hands the backward to the re-marching kernel, and the gradients are still the oracle's."""
    from dmesh_renderer_amd import _C
    B, H, W = 1, 208, 304
    gc, gd = upstream_grads(B, H, W)
    seen = []
    for opacity in ((0.85, 0.95), (0.85, 0.95), (0.02, 0.2), (0.02, 0.2)):
        d = scenes.kuhn_tets(7, B, H, W, seed=0, opacity=opacity)
        sc = oracle.scene_from_module_inputs(d, H, W)
        _, _, _, ost = oracle.tet_forward(sc)
        og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
        args = c_args(d, hip_device, tet=True)
        out = _C.render_tets(*args, H, W, 0)
        g = _C.render_tets_backward(*args, gc.to(hip_device), gd.to(hip_device), *out[3:7])
        th.cuda.synchronize()
        seen.append(_seq_state(_C, args, out[3:7], H, W))
        assert seen[-1][0] == int(ost.get("n_contrib").max())
        for got, key in zip(g, ("verts_color", "faces_opacity")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, (opacity, key)
    assert seen[0][1] == 0 and 0 < seen[1][0] <= seen[1][1], seen   # no estimate, then a complete sequence
    assert seen[2][0] > seen[2][1] > 0, seen                         # long marches after short ones: overflow -> re-march
    assert 0 < seen[3][0] <= seen[3][1], seen                        # ... and the estimate has caught up


def test_module_autograd(oracle, hip_device):
    import dmesh_renderer_amd as dmr
    d, B, H, W = _scene("small")
    dev = hip_device
    t = {k: v.to(dev) for k, v in d.items()}
    vc = t["verts_color"].clone().requires_grad_(True)
    fo = t["faces_opacity"].clone().requires_grad_(True)
    r = dmr.TetRenderer(dmr.TetRenderSettings(H, W, t["bg"], 0))
    color, depth, active = r(t["verts"].double(), t["faces"].long(), vc, fo, t["mv_mats"], t["proj_mats"],
                             t["verts_depth"], t["faces_intense"], t["tets"].long(), t["face_tets"].long(),
                             t["tet_faces"].long())
    assert active.dtype == th.bool and tuple(active.shape) == (B, H, W)
    gc, gd = upstream_grads(B, H, W)
    ((color * gc.to(dev)).sum() + (depth * gd.to(dev)).sum()).backward()
    sc = oracle.scene_from_module_inputs(d, H, W)
    _, _, _, ost = oracle.tet_forward(sc)
    og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
    assert rel_err(vc.grad.cpu().numpy(), og["verts_color"]) <= GRAD_TOL
    assert rel_err(fo.grad.cpu().numpy(), og["faces_opacity"]) <= GRAD_TOL


def test_seeded_jitter(oracle, hip_device):
    """ray_random_seed > 0 (cuda_renderer/forward.cu:82-88,120-123).  Parity with cuRAND is UNPINNED: library and
    oracle draw from the same Philox-4x32-10 stream (key = seed, counter = pixel index), so they are compared with
    each other; the seed must change the image, and the backward must re-derive the forward's jittered rays."""
    from dmesh_renderer_amd import _C
    d, B, H, W = _scene("two_views_ragged")
    args = c_args(d, hip_device, tet=True)
    gc, gd = upstream_grads(B, H, W)
    imgs = {}
    for seed in (0, 7, 8):
        sc = oracle.scene_from_module_inputs(d, H, W, seed=seed)
        ocolor, odepth, oactive, ost = oracle.tet_forward(sc)
        out = _C.render_tets(*args, H, W, seed)
        th.cuda.synchronize()
        np.testing.assert_array_equal(out[2].cpu().numpy(), oactive)
        assert np.abs(out[0].cpu().numpy() - ocolor).max() <= FWD_TOL
        assert np.abs(out[1].cpu().numpy() - odepth).max() <= FWD_TOL
        og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
        g = _C.render_tets_backward(*args, gc.to(hip_device), gd.to(hip_device), *out[3:7])
        for got, key in zip(g, ("verts_color", "faces_opacity")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, (seed, key)
        imgs[seed] = out[0].cpu().numpy()
    assert np.abs(imgs[7] - imgs[0]).max() > 1e-3 and np.abs(imgs[7] - imgs[8]).max() > 1e-3


def test_empty_inputs(hip_device):
    """No faces / no tets / no vertices (Q17: the reference has no guards here): background image, inactive pixels,
    zero gradients, no out-of-bounds access."""
    from dmesh_renderer_amd import _C
    dev = hip_device
    d, B, H, W = _scene("small")
    for P, F, T in ((0, 0, 0), (d["verts"].shape[0], 0, 0)):
        dd = dict(d)
        dd["verts"] = d["verts"][:P]; dd["verts_color"] = d["verts_color"][:P]; dd["verts_depth"] = d["verts_depth"][:, :P]
        dd["faces"] = d["faces"][:F]; dd["faces_opacity"] = d["faces_opacity"][:F]; dd["faces_intense"] = d["faces_intense"][:, :F]
        dd["face_tets"] = d["face_tets"][:F]; dd["tets"] = d["tets"][:T]; dd["tet_faces"] = d["tet_faces"][:T]
        args = c_args(dd, dev, tet=True)
        out = _C.render_tets(*args, H, W, 0)
        th.cuda.synchronize()
        bg = dd["bg"].to(dev)
        assert float(out[2].abs().max()) == 0.0
        assert th.equal(out[0], bg.view(1, 3, 1, 1).expand(B, 3, H, W).contiguous()) or float(out[0].abs().max()) == 0.0
        g = _C.render_tets_backward(*args, th.ones(B, 3, H, W, device=dev), th.ones(B, 1, H, W, device=dev), *out[3:7])
        th.cuda.synchronize()
        assert [tuple(x.shape) for x in g] == [(P, 3), (F,)]
        assert all(float(x.abs().sum()) == 0.0 for x in g)


def test_malformed_tets_stop_the_march_like_the_reference(oracle, hip_device):
    """tet_faces with a face listed twice in a tet, and with a face that is not the one its neighbour points back with: the
    reference's march stops there (`cnt != 3`, forward.cu:716-722; "error cases").  The HIP march decides that from one bit per
    face entry set when the records are built (TET_FACE_DUP, dmr_tet.hip) -- the topology of every pixel's march must still be
    the reference's, for the forward and for both backward paths."""
    from dmesh_renderer_amd import _C
    m, B, H, W = 5, 1, 112, 128
    d = scenes.kuhn_tets(m, B, H, W, seed=3, opacity=(0.05, 0.4))
    tf = d["tet_faces"].clone()
    T = tf.shape[0]
    tf[3::11, 1] = tf[3::11, 0]           # the same face in two slots of a tet
    tf[5::13, 2] = tf[(7 + 5) % T, 0]     # a face of some other tet in the third slot
    d["tet_faces"] = tf.contiguous()
    sc = oracle.scene_from_module_inputs(d, H, W)
    ocolor, odepth, oactive, ost = oracle.tet_forward(sc)
    args = c_args(d, hip_device, tet=True)
    gc, gd = upstream_grads(B, H, W)
    og = oracle.tet_backward(sc, ost, gc.numpy(), gd.numpy())
    for call in range(2):  # the second call's backward runs on the forward's march sequence
        out = _C.render_tets(*args, H, W, 0)
        th.cuda.synchronize()
        bufs = out[3:7]
        ex = lambda name, dtype: _C.export(name, args, True, ost.num_rendered, bufs, H, W, dtype).cpu().numpy()
        for name in ("first_face", "first_tet", "last_face", "last_tet"):
            np.testing.assert_array_equal(ex(name, th.int32), ost.get(name), err_msg=name)
        np.testing.assert_array_equal(ex("n_contrib", th.int32).view(np.uint32), ost.get("n_contrib"))
        np.testing.assert_array_equal(out[2].cpu().numpy(), oactive)
        assert np.abs(out[0].cpu().numpy() - ocolor).max() <= FWD_TOL
        g = _C.render_tets_backward(*args, gc.to(hip_device), gd.to(hip_device), *bufs)
        th.cuda.synchronize()
        for got, key in zip(g, ("verts_color", "faces_opacity")):
            assert rel_err(got.cpu().numpy(), og[key]) <= GRAD_TOL, (call, key)
    stopped = (oactive == 0) & (ost.get("first_face").reshape(oactive.shape) >= 0)
    assert stopped.any(), "the scene must have marches that stop inside the mesh"
