"""HIP path, through the drop-in Modules, against the committed golden fixtures
(tests/golden/wrapper_*.npz: REFERENCE Python wrapper + CPU oracle on the same seeded scenes)."""
import os

import numpy as np
import pytest
import torch as th

from dmesh_renderer_amd import scenes
from util import rel_err
import test_wrapper_cpu as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


def test_tri_golden(hip_device):
    import dmesh_renderer_amd as dmr
    g = np.load(os.path.join(GOLD, "wrapper_tri.npz"))
    d = _to(T._tri_scene(), hip_device)
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")}
    r = dmr.TriRenderer(dmr.TriRenderSettings(T.TRI["H"], T.TRI["W"], d["bg"]))
    color, depth = r(leaves["verts"], d["faces"].long(), leaves["verts_color"], leaves["faces_opacity"],
                     d["mv_mats"], d["proj_mats"], leaves["verts_depth"], leaves["faces_intense"])
    gc, gd = scenes.upstream_grads(T.TRI["B"], T.TRI["H"], T.TRI["W"])
    ((color * gc.to(hip_device)).sum() + (depth * gd.to(hip_device)).sum()).backward()
    assert np.abs(color.detach().cpu().numpy() - g["color"]).max() <= 1e-5
    assert np.abs(depth.detach().cpu().numpy() - g["depth"]).max() <= 1e-5
    for k, v in leaves.items():
        assert rel_err(v.grad.cpu().numpy(), g["grad_" + k]) <= 1e-4, k


def test_tet_golden(hip_device):
    import dmesh_renderer_amd as dmr
    g = np.load(os.path.join(GOLD, "wrapper_tet.npz"))
    d = _to(T._tet_scene(), hip_device)
    vc = d["verts_color"].clone().requires_grad_(True); fo = d["faces_opacity"].clone().requires_grad_(True)
    r = dmr.TetRenderer(dmr.TetRenderSettings(T.TET["H"], T.TET["W"], d["bg"], 0))
    color, depth, active = r(d["verts"].double(), d["faces"].long(), vc, fo, d["mv_mats"].double(), d["proj_mats"],
                             d["verts_depth"], d["faces_intense"], d["tets"].long(), d["face_tets"].long(), d["tet_faces"].long())
    gc, gd = scenes.upstream_grads(T.TET["B"], T.TET["H"], T.TET["W"])
    ((color * gc.to(hip_device)).sum() + (depth * gd.to(hip_device)).sum()).backward()
    np.testing.assert_array_equal(active.cpu().numpy(), g["active"])
    assert np.abs(color.detach().cpu().numpy() - g["color"]).max() <= 1e-5
    assert np.abs(depth.detach().cpu().numpy() - g["depth"]).max() <= 1e-5
    assert rel_err(vc.grad.cpu().numpy(), g["grad_verts_color"]) <= 1e-4
    assert rel_err(fo.grad.cpu().numpy(), g["grad_faces_opacity"]) <= 1e-4
