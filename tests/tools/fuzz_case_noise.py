"""Test infrastructure.  One case of `fuzz_campaign.py --big` again, with the float-noise picture of its dL_dverts:

    python tests/tools/fuzz_case_noise.py 180377

prints max |x - y| / max |y| for (library, oracle float), (oracle float, oracle double), (library, oracle double) -- see grad_noise.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch as th
from dmesh_renderer_amd import _C, scenes
from dmesh_renderer_amd.scenes import c_args, upstream_grads
from oracle import oracle as O

sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_fuzz_gpu import big_case
seed = int(sys.argv[1]); dev = th.device("cuda:0"); O.build()
d, B, H, W, rows = big_case(seed)
gc, gd = upstream_grads(B, H, W)
sc = O.scene_from_module_inputs(d, H, W, rows=rows)
oc, od, ost = O.tri_forward(sc)
g32 = O.tri_backward(sc, ost, gc.numpy(), gd.numpy())["verts"].astype(np.float64)
g64 = O.tri_backward(sc, ost, gc.numpy(), gd.numpy(), verts_grad_f64=True)["verts"].astype(np.float64)
args = c_args(d, dev)
out = _C.render_tris(*args, H, W, rows=rows)
hip = _C.render_tris_backward(*args, gc.to(dev), gd.to(dev), out[0], *out[3:7], rows=rows)[0].cpu().numpy().astype(np.float64)
f = np.isfinite(g32) & np.isfinite(g64)
m = max(1.0, np.abs(np.where(f, g64, 0)).max())
dist = lambda a, b: np.abs(np.where(f, a - b, 0)).max() / m
print(f"seed {seed}: B {B} H {H} W {W} F {d['faces'].shape[0]} rows {rows}  max |dL_dverts| {m:.3e}")
print(f"library vs oracle float   {dist(hip, g32):.3e}\noracle float vs double    {dist(g32, g64):.3e}\nlibrary vs oracle double  {dist(hip, g64):.3e}")
