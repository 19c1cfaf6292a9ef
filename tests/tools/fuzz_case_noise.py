"""Test infrastructure.  One case of `fuzz_campaign.py --big` again, with the float-noise picture of its dL_dverts:

    python tests/tools/fuzz_case_noise.py 180377

prints max |x - y| / max |y| for (library, oracle float), (oracle float, oracle double), (library, oracle double) -- see grad_noise.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch as th
from dmesh_renderer_amd import _C, scenes
from dmesh_renderer_amd.scenes import c_args, upstream_grads
from oracle import oracle as O

seed = int(sys.argv[1]); dev = th.device("cuda:0"); O.build()
rng = np.random.RandomState(seed)
B = int(rng.randint(1, 4)); H = int(rng.randint(17, 260)); W = int(rng.randint(17, 300))
rng.rand()  # (the campaign's tet draw)
rows = (0, 0)
H = int(rng.randint(200, 1600)); W = int(rng.randint(200, 1600))
if rng.rand() < 0.4:
    gy = (H + 15) // 16; r0 = int(rng.randint(0, gy)); rows = (r0, int(rng.randint(r0 + 1, gy + 1)))
gc, gd = upstream_grads(B, H, W)
L = int(rng.randint(1, 13)); n = int(rng.randint(20, 131))
d = scenes.layered_sheets(L, n, B, H, W, seed=seed, opacity=(0.05, float(rng.uniform(0.2, 0.95))))
if rng.rand() < 0.5: d["verts"] = d["verts"] * float(rng.uniform(0.3, 3.0))
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
