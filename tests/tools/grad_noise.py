"""Test infrastructure.  How much of dL_dverts is float rounding noise of the reference's own arithmetic?

The reference evaluates ray_tri_intersection_grad (auxiliary.h:288-333) per (pixel, face) pair in float; its cross(T, d)
-- T = eye - p0 and the ray direction d are nearly parallel for a pixel next to p0 -- loses 1e2..1e4 in relative precision.
The oracle restates that order; the HIP library (k_tri_backward_hits) sums the ray moments the gradient is linear in and
takes the cross products once per list entry.  Both are float evaluations of the same formula; this tool compares each
with the formula evaluated in double per pair (oracle, verts_grad_f64=True):

    python tests/tools/grad_noise.py C4     ->  max |x - f64| / max |f64|  for x = oracle (float) and x = HIP
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch as th
from dmesh_renderer_amd import _C, scenes
from oracle import oracle as O

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[cfgname]; d = scenes.make(cfgname); dev = th.device("cuda:0")
B, H, W = cfg.B, cfg.H, cfg.W
args = scenes.c_args(d, dev); gc, gd = scenes.upstream_grads(B, H, W)
O.build()
sc = O.scene_from_module_inputs(d, H, W)
oc, od, ost = O.tri_forward(sc)
g32 = O.tri_backward(sc, ost, gc.numpy(), gd.numpy())["verts"].astype(np.float64)
g64 = O.tri_backward(sc, ost, gc.numpy(), gd.numpy(), verts_grad_f64=True)["verts"].astype(np.float64)
o = _C.render_tris(*args, H, W)
hip = _C.render_tris_backward(*args, gc.to(dev), gd.to(dev), o[0], *o[3:7])[0].cpu().numpy().astype(np.float64)
m = max(1.0, np.abs(g64).max())
def line(name, a, b):
    e = np.abs(a - b); q = e / (2e-6 * m + 1e-3 * np.abs(b))
    print(f"{name:28s} max |diff| / max |ref| = {e.max() / m:.3e}   rms = {np.sqrt((e ** 2).mean()) / m:.3e}   "
          f"entries beyond 2e-6 max + 1e-3 |ref|: {(q > 1).sum()}   beyond 1e-5 max + 1e-3 |ref|: {(e > 1e-5 * m + 1e-3 * np.abs(b)).sum()}")
print(f"{cfgname}: dL_dverts, max |f64| = {m:.4e}")
line("oracle float  vs  double", g32, g64)
line("HIP           vs  double", hip, g64)
line("HIP           vs  oracle float", hip, g32)
