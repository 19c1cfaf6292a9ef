"""Test infrastructure. Reproduces one triangle-soup case of fuzz_campaign.py and prints the worst gradient entries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch as th
from dmesh_renderer_amd import _C
from util import c_args, upstream_grads
from oracle import oracle as O
from test_fuzz_gpu import _soup, NAMES
seed = int(sys.argv[1]); O.build(); O.lib(); dev = th.device("cuda:0")
rng = np.random.RandomState(seed)
B = int(rng.randint(1, 4)); H = int(rng.randint(17, 260)); W = int(rng.randint(17, 300)); tet = rng.rand() < 0.3
gc, gd = upstream_grads(B, H, W)
P = int(rng.randint(8, 600)); F = int(rng.randint(30, 3000))
d = _soup(seed, P, F, B, H, W)
if rng.rand() < 0.3: d["verts"] = d["verts"] * float(rng.uniform(0.05, 4.0))
sc = O.scene_from_module_inputs(d, H, W)
ocolor, odepth, ost = O.tri_forward(sc)
og = O.tri_backward(sc, ost, gc.numpy(), gd.numpy())
args = c_args(d, dev)
res = []
for rep in range(2):
    out = _C.render_tris(*args, H, W); th.cuda.synchronize()
    if os.environ.get("ZERO_POOL"):  # make recycled allocator blocks read as zeros
        th.cuda.empty_cache(); x = th.zeros(256 << 20, dtype=th.uint8, device=dev); th.cuda.synchronize(); del x
    g = _C.render_tris_backward(*args, gc.to(dev), gd.to(dev), out[0], *out[3:7]); th.cuda.synchronize()
    res.append([t.cpu().numpy() for t in g])
for i, k in enumerate(NAMES):
    a, r = res[0][i], og[k]
    fin = np.isfinite(r)
    diff = np.abs(np.where(fin, a - r, 0.0)); scale = max(1.0, float(np.abs(np.where(fin, r, 0.0)).max()))
    j = np.unravel_index(np.argmax(diff), diff.shape)
    rep_diff = float(np.abs(np.where(fin, res[0][i] - res[1][i], 0.0)).max())
    print(f"{k:14s} max|ref| {scale:.4g} max abs diff {diff.max():.4g} (rel {diff.max()/scale:.3g}) at {j}: gpu {a[j]:.9g} ref {r[j]:.9g}; gpu run-to-run diff {rep_diff:.3g}; nonfinite ref {int((~fin).sum())}")
print("B H W P F", B, H, W, P, F, "R", out[0])

if len(sys.argv) > 2:  # bisect the upstream gradient over pixels for the face named on the command line
    face = int(sys.argv[2])
    gcn, gdn = gc.numpy(), gd.numpy()
    out = _C.render_tris(*args, H, W); th.cuda.synchronize()
    def disc(mask):
        mc = (gcn * mask[:, None]).astype(np.float32); md = (gdn.reshape(B, H, W) * mask).reshape(gdn.shape).astype(np.float32)
        r = O.tri_backward(sc, ost, mc, md)["faces_opacity"][face]
        g = _C.render_tris_backward(*args, th.from_numpy(mc).to(dev), th.from_numpy(md).to(dev), out[0], *out[3:7])[2][face].item()
        return g, r
    lo, hi = 0, B * H * W
    while hi - lo > 1:
        mid = (lo + hi) // 2
        m = np.zeros(B * H * W, np.float32); m[lo:mid] = 1
        g, r = disc(m.reshape(B, H, W))
        if abs(g - r) > 1e-3 * max(1, abs(r)): hi = mid
        else: lo = mid
    m = np.zeros(B * H * W, np.float32); m[lo] = 1
    g, r = disc(m.reshape(B, H, W))
    b_, y_, x_ = np.unravel_index(lo, (B, H, W))
    nc = ost.get("n_contrib").reshape(B, H, W)[b_, y_, x_]
    print(f"pixel {lo} (b {b_}, y {y_}, x {x_}): faces_opacity[{face}] gpu {g:.7g} ref {r:.7g}; n_contrib {nc}; tile ({x_ // 16}, {y_ // 16})")
    fl = ost.get("values"); rg = ost.get("ranges").reshape(-1, 2)
    t = (b_ * ((H + 15) // 16) + y_ // 16) * ((W + 15) // 16) + x_ // 16
    lst = fl[rg[t, 0]:rg[t, 1]]
    pos = np.nonzero(lst == face)[0]
    print("tile list length", len(lst), "position of the face in the list", pos, "opacity", float(d["faces_opacity"][face]), "faces[face]", d["faces"][face].tolist())
    print("opacities of the faces before it (first 1.0 at):", np.nonzero(d["faces_opacity"].numpy()[lst[:int(nc)]] == 1.0)[0][:5])
