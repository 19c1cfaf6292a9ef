"""Test infrastructure. Reproduces one tet case of fuzz_campaign.py and prints where the index images differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch as th
from dmesh_renderer_amd import _C
from util import c_args
from oracle import oracle as O
from test_fuzz_gpu import _delaunay
seed = int(sys.argv[1]); O.build(); O.lib(); dev = th.device("cuda:0")
rng = np.random.RandomState(seed)
B = int(rng.randint(1, 4)); H = int(rng.randint(17, 260)); W = int(rng.randint(17, 300)); tet = rng.rand() < 0.3
npts = int(rng.randint(20, 500))
d = _delaunay(seed, npts, B, H, W)
sc = O.scene_from_module_inputs(d, H, W)
ocolor, odepth, oactive, ost = O.tet_forward(sc)
args = c_args(d, dev, tet=True)
out = _C.render_tets(*args, H, W, 0); th.cuda.synchronize()
ex = lambda n, dt: _C.export(n, args, True, 0, out[3:7], H, W, dt).cpu().numpy()
for name in ("first_face", "first_tet", "last_face", "n_contrib"):
    try:
        g = ex(name, th.int32); o = ost.get(name)
    except Exception as e:
        print(name, "n/a", e); continue
    g = g.reshape(o.shape) if g.size == o.size else g
    diff = np.argwhere(g.reshape(-1) != o.reshape(-1).astype(g.dtype))
    print(name, "mismatches", len(diff), [(int(i), int(g.reshape(-1)[i]), int(o.reshape(-1)[i])) for i in diff[:8, 0]])
a = out[2].cpu().numpy(); diff = np.argwhere(a.reshape(-1) != oactive.reshape(-1))
print("active mismatches", len(diff), diff[:8, 0].tolist())
print("B H W npts", B, H, W, npts, "T", d["tets"].shape[0], "F", d["faces"].shape[0])
i = int(sys.argv[2]) if len(sys.argv) > 2 else None
if i is not None:
    for name in ("final_T", "final_prev_T"):
        g = ex(name, th.float32).reshape(-1); o = ost.get(name).reshape(-1)
        print(name, "gpu %.9g oracle %.9g" % (g[i], o[i]), "exp: gpu-side value %.9g, oracle-side value %.9g (T_EPS 1e-4)" % (np.exp(np.float64(g[i])), np.exp(np.float64(o[i]))))
