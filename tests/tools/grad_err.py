"""Test infrastructure. Per-tensor forward/gradient error of the tri path against the CPU oracle (debug aid)."""
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
oc, od, ost = O.tri_forward(sc); og = O.tri_backward(sc, ost, gc.numpy(), gd.numpy())
for it in range(2):
    o = _C.render_tris(*args, H, W)
    g = _C.render_tris_backward(*args, gc.to(dev), gd.to(dev), o[0], *o[3:7])
    print("fwd err", float(np.abs(o[1].cpu().numpy() - oc).max()), float(np.abs(o[2].cpu().numpy() - od).max()))
    for t, k in zip(g, ("verts", "verts_color", "faces_opacity", "verts_depth", "faces_intense")):
        a, r = t.cpu().numpy().astype(np.float64), og[k].astype(np.float64)
        e = np.abs(a - r); i = np.unravel_index(np.argmax(e), e.shape)
        print(f"{k:14s} rel {e.max() / max(1, np.abs(r).max()):.3e} maxabs ref {np.abs(r).max():.3e} worst at {i}: got {a[i]:.6e} ref {r[i]:.6e}; "
              f"n(|err|>1e-4*max) = {(e > 1e-4 * max(1, np.abs(r).max())).sum()} nan {np.isnan(a).sum()}")
        lim = 2e-6 * max(1, np.abs(r).max()) + 1e-3 * np.abs(r); q = e / lim; j = np.unravel_index(np.argmax(q), q.shape)
        print(f"{'':14s} per entry (2e-6 max + 1e-3 |ref|): worst ratio {q.max():.3f} at {j}: got {a[j]:.6e} ref {r[j]:.6e}; n(ratio > 1) = {(q > 1).sum()}"
              f"; n(ratio > 0.5) = {(q > 0.5).sum()}")
